#!/usr/bin/env python3
"""tools/depth_ab.py — per-step cost of the pass depths on one box: for each grid, fresh steppers with
"fuse" = 5, 6, 7 (balanced passes of that depth) tune their chunk height and time 3 x 840 steps; best of
three, depths interleaved over three rounds so that clock drift cancels.  Feeds STEP_COST in csrc/api.cpp."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

csim = load_package()
csim.lib()
csim.set_device(0)
for (nx, ny) in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(16384, 16384), (8192, 8192), (4096, 8192)]:
    res = {}
    for rnd in range(3):
        for depth in (2, 3, 4, 5, 6, 7):
            st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("dddd"))
            st.set_option("fuse", depth)
            st.init_gaussian()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                st.run(0.05, 0.1, 0.5, 0.25, 84)
                st.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                st.run(0.05, 0.1, 0.5, 0.25, 840)
                st.sync()
                best = min(best, time.perf_counter() - t0)
            rows = st.get_option("tuned_rows")
            st.close()
            res.setdefault(depth, []).append((round(nx * ny * 840 / best / 1e6), rows))
    print(json.dumps(dict(grid=f"{nx}x{ny}", **{f"fuse_{d}": res[d] for d in res})), flush=True)
