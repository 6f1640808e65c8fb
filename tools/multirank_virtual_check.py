#!/usr/bin/env python3
"""tools/multirank_virtual_check.py WORLD [DEPTHS...] — every golden case with WORLD-rank data through N virtual ranks in
this process (tests/virtual_ranks.py), one line per case and depth as it finishes (a hang shows where)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from test_multirank_gloo import cases_with  # noqa: E402
from virtual_ranks import VirtualRanks, tile_mask  # noqa: E402

csim = load_package()
csim.lib()
csim.set_device(0)
world = int(sys.argv[1])
depths = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 7]
bad = 0
for path in cases_with(world):
    z = np.load(path, allow_pickle=False)
    m = json.loads(str(z["meta"]))
    for depth in depths:
        print(os.path.basename(path), "depth", depth, "...", end=" ", flush=True)
        vr = VirtualRanks(csim, world, m["nx"], m["ny"], m["dx"], m["dy"], csim.bc_codes(m["bc"]), fuse=depth)
        vr.upload_global(z["u0"])
        vr.advance(m["D"], float(z["dt_effective"]), m["vx"], m["vy"], m["steps"], depth=depth)
        ok = True
        for r, dec in enumerate(vr.decs):
            got, want = vr.download(r), z[f"local_np{world}_rank{r}"]
            mask = tile_mask(dec)
            ok = ok and bool(np.array_equal(got[mask], want[mask]))
        vr.close()
        print("ok" if ok else "MISMATCH", flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
