#!/usr/bin/env python3
"""tools/multirank6_check.py — 6 ranks (3x2 process grid: ranks with three neighbour sides and two
diagonal peers) of the HIP stepper on ONE GPU, faces over gloo, against the golden vectors of the
reference's `mpirun -np 6` run.  Not part of pytest: with the test runner's own GPU context the
box's cap of 6 GPU processes would be exceeded; this launcher never touches the GPU itself."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_multirank_gloo import cases_with, launch  # noqa: E402

ok = True
for engine in ("hip-external", "hip-external2", "hip-external4", "hip-external6"):
    for case in cases_with(6):
        rc, out = launch(6, engine, case, timeout=600)
        good = rc == 0 and "ok=True" in out
        print(engine, os.path.basename(case), "ok" if good else "FAILED\n" + out[-2000:], flush=True)
        ok = ok and good
sys.exit(0 if ok else 1)
