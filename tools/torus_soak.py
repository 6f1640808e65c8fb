#!/usr/bin/env python3
"""tools/torus_soak.py [STEPS] [VX VY] — long consistency run of the multi-rank schedules on the self-linked
torus (one GPU): serial exchange, frame-first (1), merged (3), bulk-first (4) and the default (5) must
leave bit-identical fields after thousands of steps (a stream-ordering race would show up as a
mismatch), in runs cut into uneven pieces so that pass depths and final passes vary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
vx, vy = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (-0.5, 0.25)   # 0 0: the diffusion-only flavour
csim = load_package()
csim.lib()
csim.set_device(0)
ok = True
for nx, ny, bc, sides in ((2048, 4096, "dddd", (1, 1, 1, 1)), (4096, 1024, "dndn", (1, 1, 0, 0)), (1000, 3000, "nnpd", (0, 0, 1, 1))):
    ref = None
    for opts in (dict(overlap=0), dict(overlap=1), dict(overlap=3), dict(overlap=4), dict(overlap=5), dict(overlap=4, fuse=7),
                 dict(overlap=1, fuse=4), dict(overlap=3, direct_faces=0), dict(overlap=3, fused_2c=0), dict(overlap=3, fuse=7),
                 dict(overlap=4, relay=0), dict(overlap=5, relay_events=1), dict(overlap=5, fuse=5)):
        d = csim.decomp_init(1, 0, nx, ny)
        for k in range(4):
            d.nbr[k] = 0 if sides[k] else csim.NO_NEIGHBOR
        st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes(bc))
        st.comm_init(csim.comm_unique_id())
        for k, v in opts.items():
            st.set_option(k, v)
        st.init_gaussian(1.0, 0.02, 0.03, 0.97)
        done, piece = 0, 1
        while done < steps:
            n = min(piece, steps - done)
            st.run(0.1, 0.1, vx, vy, n)
            done += n
            piece = piece * 3 + 1 if piece < 700 else 97
        out = st.download()
        st.close()
        if ref is None:
            ref = out
        same = bool(np.array_equal(out[1:-1, 1:-1].view(np.int64), ref[1:-1, 1:-1].view(np.int64)))   # bits: +0 != -0
        ok = ok and same
        print(nx, ny, bc, sides, opts, "identical" if same else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
