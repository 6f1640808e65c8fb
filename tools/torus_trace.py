#!/usr/bin/env python3
"""tools/torus_trace.py MODE [NXxNY] — one configuration of tools/torus_bench.py for rocprofv3
--kernel-trace (MODE = single | torus-overlap | torus-serial); tools/trace_timeline.py turns the
kernel trace into a per-pass timeline (start, duration, gap to the previous kernel)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "torus-overlap"
nx, ny = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4096x8192").split("x"))
csim = load_package()
csim.lib()
csim.set_device(0)
d = csim.decomp_init(1, 0, nx, ny)
links = os.environ.get("LINKS", "1111")   # which sides (left right bottom top) are linked to the rank itself
if mode != "single":
    for k in range(4):
        d.nbr[k] = 0 if links[k] == "1" else csim.NO_NEIGHBOR
st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes(os.environ.get("BC", "dddd")))
if mode != "single":
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", {"torus-overlap": 1, "torus-serial": 0, "torus-merged": 3, "torus-bulkfirst": 4, "torus-auto": 5}[mode])
st.init_gaussian()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    st.run(0.05, 0.1, 0.5, 0.25, 60)
    st.sync()
run = int(os.environ.get("RUN", "0"))   # steps per csim_stepper_run call (0: one call of 120 steps)
t0 = time.perf_counter()
if run > 0:
    for _ in range(120 // run):
        st.run(0.05, 0.1, 0.5, 0.25, run)
else:
    st.run(0.05, 0.1, 0.5, 0.25, 120)
st.sync()
print(mode, nx, ny, "steps per call", run or 120, "ms/step", (time.perf_counter() - t0) / 120 * 1e3)
st.close()
