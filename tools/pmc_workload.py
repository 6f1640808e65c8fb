#!/usr/bin/env python3
"""tools/pmc_workload.py — the sweep launches whose hardware counters tools/gpu_pmc.sh collects.

A fixed, printed sequence of launches on the bench grid: for each boundary mix, every kernel
instantiation bench.py can time (single-step k_sweep_dpp, k_sweepO_dpp<T = 2..7>) at the chunk heights
the on-device trial picks from (T = 5, 6, 7).  The sequence goes to stdout as one JSON line ("PMC_SEQUENCE")
so that the post-processor can attribute the k-th sweep dispatch of the profile to its configuration.
Runs under `rocprofv3 --pmc ...` (one counter set per process run)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=16384)
    ap.add_argument("--ny", type=int, default=16384)
    ap.add_argument("--bcs", nargs="+", default=["dddd", "nnnn"])
    ap.add_argument("--rows", type=int, nargs="+", default=[110, 134, 158, 182, 206, 230])
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--physics", default="0.05,0.1,0.5,0.25", help="D,dt,vx,vy (default: bench.py PHYS; vx = vy = 0: the diffusion-only flavour)")
    ap.add_argument("--depths", type=int, nargs="+", default=[1, 2, 3, 4, 5, 6, 7])
    args = ap.parse_args()
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    D, dt, vx, vy = (float(v) for v in args.physics.split(","))
    seq = []
    for bc in args.bcs:
        st = csim.Stepper.single(args.nx, args.ny, 1.0, 1.0, csim.bc_codes(bc))
        st.set_option("autotune", 0)
        st.init_gaussian(1.0, 0.05, 0.5, 0.5)
        cfgs = [(T, 0) for T in (1, 2, 3, 4) if T in args.depths] + [(T, r) for T in (5, 6, 7) if T in args.depths for r in args.rows]
        for T, ry in cfgs:
            st.set_option("fuse", 0 if T == 1 else T)
            st.set_option("rows_per_chunk", ry)
            st.run(D, dt, vx, vy, T * args.passes)
            st.sync()
            used = st.get_option("last_rows") if T > 1 else (ry or 64)
            seq.append(dict(bc=bc, steps_per_launch=T, rows_per_chunk=used, launches=args.passes))
        st.close()
    print("PMC_SEQUENCE " + json.dumps(dict(nx=args.nx, ny=args.ny, sequence=seq)), flush=True)


if __name__ == "__main__":
    main()
