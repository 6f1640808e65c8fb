#!/usr/bin/env python3
"""tools/torus_bench.py — the multi-rank code path (deep-face pack/unpack, RCCL group exchange on
the comm stream, frame tiles on the high-priority stream) timed on ONE GPU by linking a 1-rank
communicator to itself in all 8 directions.  Tile shapes are the per-GPU tiles of the 16384^2
strong-scaling run; the same tile without neighbours is timed next to it as the no-exchange bound."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", nargs="+", default=["8192x16384", "8192x8192", "4096x8192"])
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--run", type=int, default=0, help="steps per csim_stepper_run call (0 = all of --steps in one call)")
    ap.add_argument("--modes", nargs="+", default=["single", "torus-auto", "torus-merged", "torus-bulkfirst",
                                                   "torus-overlap", "torus-serial"])
    ap.add_argument("--links", default="1111", help="which sides (left right bottom top) are linked to the rank itself; the others are "
                                                    "physical edges — e.g. 1100 = the mid-x tile of a 4 x 1 row, 1101 = a mid-x tile of the 4 x 2 grid")
    ap.add_argument("--bc", default="dddd")
    ap.add_argument("--physics", default="0.05,0.1,0.5,0.25", help="D,dt,vx,vy (vx = vy = 0: the diffusion-only flavour, BASELINE configs[1])")
    args = ap.parse_args()
    phys = tuple(float(v) for v in args.physics.split(","))
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    for sh in args.shape:
        nx, ny = (int(v) for v in sh.split("x"))
        for mode in args.modes:
            d = csim.decomp_init(1, 0, nx, ny)
            if not mode.startswith("single"):
                for k in range(4):
                    d.nbr[k] = 0 if args.links[k] == "1" else csim.NO_NEIGHBOR
            st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes(args.bc))
            if mode.startswith("single"):
                for tok in mode.split("+")[1:]:
                    k, v = tok.split("=")
                    st.set_option(k, int(v))
            else:
                st.comm_init(csim.comm_unique_id())
                st.set_option("overlap", {"torus-overlap": 1, "torus-serial": 0, "torus-merged": 3, "torus-bulkfirst": 4, "torus-auto": 5}[mode.split("+")[0]])
                for tok in mode.split("+")[1:]:   # e.g. torus-merged+frame_fence=1+frame_prio=0
                    k, v = tok.split("=")
                    st.set_option(k, int(v))
            st.init_gaussian()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:  # leave the idle clocks (and let the stepper tune its chunking)
                st.run(*phys, 60)
                st.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                if args.run > 0:   # many short runs back to back, like a driver loop between snapshots
                    for _ in range(args.steps // args.run):
                        st.run(*phys, args.run)
                else:
                    st.run(*phys, args.steps)
                st.sync()
                best = min(best, time.perf_counter() - t0)
            st.close()
            print(json.dumps(dict(tile=sh, mode=mode, links=args.links, bc=args.bc, physics=args.physics, steps_per_run=args.run or args.steps, ms_per_step=best / args.steps * 1e3,
                                  mcells=nx * ny * args.steps / best / 1e6)), flush=True)


if __name__ == "__main__":
    main()
