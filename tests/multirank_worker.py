#!/usr/bin/env python3
"""tests/multirank_worker.py — one rank of an N-rank run of the hot path (launched by
tests/test_multirank_gloo.py / tests/test_gpu_multirank.py through torch.distributed.run).

The decomposition always comes from the PRODUCT (csim_decomp_init in libcsim.so); halos travel
over torch.distributed gloo point-to-point messages posted in the order of reference
src/halo.cpp:28-43.  Two engines step the local tile:
  --engine oracle        CPU: oracle/cpu_stepper.c (tests the N>1 host logic without a GPU)
  --engine hip-external  GPU: the HIP stepper with csim_stepper_halo_pack/_unpack
                         (several ranks may share one GPU; RCCL refuses that, gloo does not)
  --engine hip-external{2..7}  GPU: same, but 2..7 steps per call (one fused pass) with faces
                         of that depth in 8 directions (csim_stepper_faces_pack/_unpack)
  --engine hip-rccl[N]   GPUs: one rank per GPU, halos over RCCL (exchange schedule N); only where
                         at least as many GPUs as ranks are visible
Rank 0 gathers the global interior and compares it bit-for-bit with the golden fixture."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import cpu_oracle as ora  # noqa: E402

load_package()
from climate_sim_mpi_cpp_amd.host_transport import exchange, exchange8  # noqa: E402


def edges(u, nbr):
    return [u[1:-1, 1].copy() if nbr[0] >= 0 else None, u[1:-1, -2].copy() if nbr[1] >= 0 else None,
            u[1, 1:-1].copy() if nbr[2] >= 0 else None, u[-2, 1:-1].copy() if nbr[3] >= 0 else None]


def main():
    # a rank that is stuck (rendezvous, a peer that died, a stream that never drains) says WHERE and leaves, so that
    # the launching test fails with a stack instead of sitting out its timeout
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("CSIM_WORKER_WATCHDOG_S", "150")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="oracle")
    ap.add_argument("--case", required=True, help="path of a tests/golden/run_*.npz fixture")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    csim = load_package()
    z = np.load(args.case, allow_pickle=False)
    m = json.loads(str(z["meta"]))
    dt = float(z["dt_effective"])
    dec = csim.decomp_init(world, rank, m["nx"], m["ny"])
    want_dec = z[f"decomp_np{world}"][rank]
    assert list(dec.as_dict().values()) == list(want_dec), "decomposition differs from MPI's"
    nbr = list(dec.nbr)
    phys = [1 if n < 0 else 0 for n in nbr]
    bc = csim.bc_codes(m["bc"])
    u = np.zeros((dec.ny_local + 2, dec.nx_local + 2))
    u[1:-1, 1:-1] = z["u0"][dec.y_offset:dec.y_offset + dec.ny_local,
                            dec.x_offset:dec.x_offset + dec.nx_local]

    if args.engine == "oracle":
        tmp = u.copy()
        for _ in range(m["steps"]):
            got = exchange(edges(u, nbr), nbr)
            if got[0] is not None:
                u[1:-1, 0] = got[0]
            if got[1] is not None:
                u[1:-1, -1] = got[1]
            if got[2] is not None:
                u[0, 1:-1] = got[2]
            if got[3] is not None:
                u[-1, 1:-1] = got[3]
            ora.step_tile(u, tmp, m["dx"], m["dy"], m["D"], m["vx"], m["vy"], dt, bc, phys)
            u, tmp = tmp, u
        local = u
    elif args.engine == "hip-external":
        csim.lib()
        csim.set_device(0)
        st = csim.Stepper(dec, m["dx"], m["dy"], bc)
        st.set_option("external_halo", 1)
        st.upload(u)
        for _ in range(m["steps"]):
            st.halo_unpack(exchange(st.halo_pack(), nbr))
            st.run(m["D"], dt, m["vx"], m["vy"], 1)
        local = st.download()
        st.close()
    elif args.engine.startswith("hip-external") and args.engine[-1] in "234567":
        # `depth` reference steps per call (one fused HBM pass) with deep faces, then single steps
        depth = int(args.engine[-1])
        csim.lib()
        csim.set_device(0)
        st = csim.Stepper(dec, m["dx"], m["dy"], bc)
        st.set_option("external_halo", 1)
        st.set_option("fuse", depth)         # (auto mode offers 6, the depth that is cheapest per step)
        st.upload(u)
        depth = min(depth, st.fuse_limit())  # tiny tiles cap the depth (same value on every rank)
        remaining = m["steps"]
        while remaining >= 3 and depth >= 2:
            t = min(depth, remaining - 1)
            peers, _ = st.faces_neighbors(t)
            st.faces_unpack(t, exchange8(st.faces_pack(t), peers))
            st.run(m["D"], dt, m["vx"], m["vy"], t)
            remaining -= t
        while remaining > 0:
            st.halo_unpack(exchange(st.halo_pack(), nbr))
            st.run(m["D"], dt, m["vx"], m["vy"], 1)
            remaining -= 1
        local = st.download()
        st.close()
    elif args.engine.startswith("hip-rccl"):
        # one rank per GPU, halos over RCCL point-to-point (needs as many visible GPUs as ranks); the suffix
        # selects the exchange schedule ("hip-rccl" = the stepper's default, "hip-rccl3" = overlap 3, ...)
        csim.lib()
        ndev = csim.device_count()
        assert ndev >= world, f"{world} ranks need {world} GPUs, {ndev} visible"
        csim.set_device(rank)
        st = csim.Stepper(dec, m["dx"], m["dy"], bc)
        box = [csim.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        st.comm_init(box[0])
        if args.engine[-1].isdigit():
            st.set_option("overlap", int(args.engine[-1]))
        st.upload(u)
        st.run(m["D"], dt, m["vx"], m["vy"], 1)                    # uneven calls: single step, fused passes
        st.run(m["D"], dt, m["vx"], m["vy"], m["steps"] - 1)
        local = st.download()
        st.close()
    else:
        raise SystemExit("unknown engine")

    # per-rank full local array (ghosts included, corners excluded: SURVEY Q7)
    want_local = z[f"local_np{world}_rank{rank}"]
    mask = np.ones(local.shape, bool)
    if world > 1:
        mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    ok_local = bool(np.array_equal(local[mask], want_local[mask]))
    parts = [None] * world
    dist.all_gather_object(parts, (dec.x_offset, dec.y_offset, local[1:-1, 1:-1].copy(), ok_local))
    ok = True
    if rank == 0:
        glob = np.zeros((m["ny"], m["nx"]))
        for xo, yo, a, okl in parts:
            glob[yo:yo + a.shape[0], xo:xo + a.shape[1]] = a
            ok = ok and okl
        ok = ok and bool(np.array_equal(glob, z["u_final"]))
        print(f"MULTIRANK engine={args.engine} world={world} case={os.path.basename(args.case)} ok={ok}",
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
