import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from oracle import cpu_oracle as ora
csim = load_package(); csim.lib(); csim.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ny = int(sys.argv[2]) if len(sys.argv) > 2 else n
steps = 6
bc = "pppp"
rng = np.random.default_rng(42)
u0 = np.zeros((ny + 2, n + 2)); u0[1:-1, 1:-1] = rng.random((ny, n))
want = u0.copy(); ora.run_single(want, 1.0, 1.0, 1.0, 0.0, 0.0, 0.1, ora.bc_codes(bc), steps)
for opts in [dict(fuse=0), dict(fuse=1), dict(variant=2, fuse=0), dict(variant=3, fuse=0)]:
    st = csim.Stepper.single(n, ny, 1.0, 1.0, csim.bc_codes(bc))
    for k, v in opts.items(): st.set_option(k, v)
    st.upload(u0); st.run(1.0, 0.1, 0.0, 0.0, steps); got = st.download(); st.close()
    bad = np.argwhere(got != want)
    print(opts, "mismatches:", len(bad), "rows", (bad[:,0].min(), bad[:,0].max()) if len(bad) else None,
          "cols", (bad[:,1].min(), bad[:,1].max()) if len(bad) else None, flush=True)
    if len(bad):
        rows = np.unique(bad[:,0]); print("  distinct rows:", rows[:20], "... n=", len(rows))
        cols = np.unique(bad[:,1]); print("  distinct cols:", cols[:20], "... n=", len(cols))
