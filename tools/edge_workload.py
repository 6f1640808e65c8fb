#!/usr/bin/env python3
"""tools/edge_workload.py — launches whose counters separate the EDGE body of k_sweepO_dpp from the interior body.

A 224-column tile is two strips at T = 7 (and at T = 6): the first one holds the left ghost column, the second the
right one, so EVERY wavefront of a single-rank launch runs the edge body with exactly one physical side — and the
very same tile run as a rank whose four sides have neighbours (external-halo mode, faces of zeros) runs the interior
body on every wavefront.  rocprofv3 --pmc SQ_INSTS_VALU (+ --kernel-trace for the durations) over this script gives
the executed-instruction ratio edge / interior per boundary kind; tools/edge_collect.py reads it.
Prints the launch sequence as one JSON line (EDGE_SEQUENCE)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=224)
    ap.add_argument("--ny", type=int, default=131072)
    ap.add_argument("--depths", type=int, nargs="+", default=[7, 6])
    ap.add_argument("--passes", type=int, default=4)
    args = ap.parse_args()
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    D, dt, vx, vy = 0.05, 0.1, 0.5, 0.25
    seq = []
    rng = np.random.default_rng(1)
    u0 = np.zeros((args.ny + 2, args.nx + 2))
    u0[1:-1, 1:-1] = rng.random((args.ny, args.nx))
    for T in args.depths:
        # interior body on every wavefront: all four sides have "neighbours" (the rank itself), faces carried by us
        dec = csim.decomp_init(1, 0, args.nx, args.ny)
        for k in range(4):
            dec.nbr[k] = 0
        st = csim.Stepper(dec, 1.0, 1.0, csim.bc_codes("dddd"))
        st.set_option("external_halo", 1)
        st.set_option("fuse", T)
        st.set_option("autotune", 0)
        st.set_option("rows_per_chunk", 182 if T == 7 else 182 + 2)
        st.upload(u0)
        for _ in range(args.passes):
            st.faces_unpack(T, st.faces_pack(T))
            st.run(D, dt, vx, vy, T)
            st.sync()
        seq.append(dict(body="interior", bc="torus", T=T, launches=args.passes, rows=st.get_option("last_rows")))
        st.close()
        for bc in ("dddd", "nnnn", "pppp", "dnnd"):
            st = csim.Stepper.single(args.nx, args.ny, 1.0, 1.0, csim.bc_codes(bc))
            st.set_option("fuse", T)
            st.set_option("autotune", 0)
            st.set_option("rows_per_chunk", 182 if T == 7 else 182 + 2)
            st.upload(u0)
            st.run(D, dt, vx, vy, T * args.passes + T)     # the last pass of a run also emits the FinLines: counted apart
            st.sync()
            seq.append(dict(body="edge", bc=bc, T=T, launches=args.passes, rows=st.get_option("last_rows")))
            seq.append(dict(body="edge+fin", bc=bc, T=T, launches=1, rows=st.get_option("last_rows")))
            st.close()
    print("EDGE_SEQUENCE " + json.dumps(dict(nx=args.nx, ny=args.ny, sequence=seq)), flush=True)


if __name__ == "__main__":
    main()
