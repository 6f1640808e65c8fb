#!/usr/bin/env python3
"""tools/multirank_check.py — N ranks of the stepper (one process each, RCCL halos) against the
single-tile oracle on the same global field.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port 29517 tools/multirank_check.py [--same-gpu] [--nx 512 --ny 384 --steps 9]
--same-gpu puts every rank on device 0 (only useful where RCCL tolerates that)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--ny", type=int, default=384)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--bc", default="dnpd")
    ap.add_argument("--overlap", type=int, default=1)
    args = ap.parse_args()
    import torch.distributed as dist
    from __graft_entry__ import load_package
    from oracle import cpu_oracle as ora
    csim = load_package()
    csim.lib()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    csim.set_device(0 if args.same_gpu else int(os.environ["LOCAL_RANK"]))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(99)
    g0 = rng.standard_normal((args.ny, args.nx))
    dec = csim.decomp_init(world, rank, args.nx, args.ny)
    st = csim.Stepper(dec, 1.0, 1.0, csim.bc_codes(args.bc))
    box = [csim.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    st.comm_init(box[0])
    st.set_option("overlap", args.overlap)
    loc = np.zeros((dec.ny_local + 2, dec.nx_local + 2))
    loc[1:-1, 1:-1] = g0[dec.y_offset:dec.y_offset + dec.ny_local, dec.x_offset:dec.x_offset + dec.nx_local]
    st.upload(loc)
    st.run(D, dt, vx, vy, 2)
    st.run(D, dt, vx, vy, args.steps - 2)
    mine = st.download_interior()
    st.close()
    parts = [None] * world
    dist.all_gather_object(parts, (dec.x_offset, dec.y_offset, mine))
    ok = True
    if rank == 0:
        got = np.zeros_like(g0)
        for xo, yo, a in parts:
            got[yo:yo + a.shape[0], xo:xo + a.shape[1]] = a
        want = np.zeros((args.ny + 2, args.nx + 2))
        want[1:-1, 1:-1] = g0
        ora.run_single(want, 1.0, 1.0, D, vx, vy, dt, ora.bc_codes(args.bc), args.steps)
        ok = np.array_equal(got, want[1:-1, 1:-1])
        print(f"multirank_check world={world} dims={dec.dims[0]}x{dec.dims[1]} overlap={args.overlap} "
              f"bit_exact={ok} linf={np.abs(got - want[1:-1, 1:-1]).max()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
