#!/usr/bin/env python3
"""tools/sweep_variants.py — interleaved A/B timing of the fused-sweep kernel variants in ONE
process on one GPU (cdna guide §5.4 rule 24).  Prints one line per configuration with the
HIP-event kernel time and the algorithmic GB/s (16 B per cell update)."""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[16384])
    ap.add_argument("--shape", type=str, nargs="*", default=[], help="NXxNY tiles, e.g. 4096x8192")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--ry", type=int, nargs="+", default=[32, 64, 128, 256])
    ap.add_argument("--pf", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--swz", type=int, nargs="+", default=[1])
    ap.add_argument("--fuse", type=int, nargs="+", default=[0, 2, 3, 4])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    results = []
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in args.shape] or [(n, n) for n in args.n]
    for (n, n_y) in shapes:
        st = csim.Stepper.single(n, n_y, 1.0, 1.0, csim.bc_codes("dddd"))
        st.init_gaussian()
        st.set_option("profile", 1)
        cfgs = []
        for v in args.variants:
            if v in (0, 1):
                cfgs += [dict(variant=v, rows_per_chunk=r, prefetch=p, xcd_swizzle=s, fuse=f)
                         for r, p, s, f in itertools.product(args.ry, args.pf, args.swz, args.fuse)]
            elif v == 2:
                cfgs += [dict(variant=2, rows_per_chunk=r, prefetch=0, xcd_swizzle=s, fuse=0)
                         for r, s in itertools.product(args.ry, args.swz)]
            else:
                cfgs += [dict(variant=3, rows_per_chunk=0, prefetch=0, xcd_swizzle=1, fuse=0)]
        best = {}
        for rnd in range(args.rounds):
            for ci, cfg in enumerate(cfgs):
                for k, val in cfg.items():
                    st.set_option(k, val)
                st.run(0.05, 0.1, 0.5, 0.25, 5)
                st.sync()
                st.reset_timers()
                st.run(0.05, 0.1, 0.5, 0.25, args.steps)
                ms, cnt, nst = st.kernel_time()
                per = ms / nst  # ms per time step
                best.setdefault(ci, []).append(per)
        for ci, cfg in enumerate(cfgs):
            ts = sorted(best[ci])
            med = ts[len(ts) // 2]
            gbs = n * n_y * 16.0 / (med * 1e-3) / 1e9
            rec = dict(n=n, ny=n_y, **cfg, ms_med=med, ms_min=ts[0], gbs_med=gbs,
                       mcells=n * n_y / med / 1e3)
            results.append(rec)
            print(json.dumps(rec), flush=True)
        st.close()
    if args.out:
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
