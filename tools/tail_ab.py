#!/usr/bin/env python3
"""tools/tail_ab.py — A/B of the tail regions of a whole-field launch on one box: for each grid and each
tail_split mode (0 off, 1 half + quarter height, 2 half height only) a fresh stepper tunes its chunk height
and times 3 x 600 steps; best of three, modes interleaved twice so that clock drift cancels."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

csim = load_package()
csim.lib()
csim.set_device(0)
for (nx, ny) in [(16384, 16384), (8192, 8192), (8192, 16384), (4096, 4096)]:
    res = {}
    for rnd in range(2):
        for mode in (0, 1, 2):
            st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("dddd"))
            st.set_option("tail_split", mode)
            st.init_gaussian()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                st.run(0.05, 0.1, 0.5, 0.25, 60)
                st.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                st.run(0.05, 0.1, 0.5, 0.25, 600)
                st.sync()
                best = min(best, time.perf_counter() - t0)
            rows = st.get_option("tuned_rows")
            st.close()
            res.setdefault(mode, []).append((nx * ny * 600 / best / 1e6, rows))
    print(json.dumps(dict(grid=f"{nx}x{ny}", **{f"tail_split_{m}": [(round(v), r) for v, r in res[m]] for m in res})), flush=True)
