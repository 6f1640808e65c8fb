#!/usr/bin/env python3
"""tests/multirank_fullsize_worker.py — one rank of a FULL-SIZE decomposed run of the hot path on a
shared GPU (launched by tests/test_gpu_multirank.py): BASELINE configs[3] (16384^2 on 2 x 2 ranks) and the
grid of configs[4] (32768^2, all-Neumann) as far as one card can host them.

All ranks share device 0, so the faces travel host-staged over gloo (RCCL refuses two ranks on one device);
everything else is the product's multi-rank path: csim_decomp_init, deep faces in 8 directions
(csim_stepper_faces_pack / _unpack), ghost fill with its halo extension, fused passes.  The reference is
decomposition-invariant, so every rank checks its tile — ghost lines of its physical sides included — bit
for bit against the same region of a single-rank run of the whole grid, which it makes itself on the same
GPU (and which tests/test_gpu_parity.py's full-field tests tie to the oracle)."""
import argparse
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from climate_sim_mpi_cpp_amd.host_transport import advance  # noqa: E402


def field_rows(nxg, nyg, y0, ny, x0, nx):
    """deterministic initial interior, computable per tile: a hotspot plus a cell-wise pattern that makes
    every cell rounding-sensitive"""
    j = (np.arange(y0, y0 + ny, dtype=np.float64) + 0.5)[:, None]
    i = (np.arange(x0, x0 + nx, dtype=np.float64) + 0.5)[None, :]
    r2 = (i - 0.5 * nxg) ** 2 + (j - 0.5 * nyg) ** 2
    sig = 0.05 * min(nxg, nyg)
    return np.exp(-r2 / (2 * sig * sig)) + 0.05 * np.sin(0.37 * i) * np.cos(0.11 * j) + 0.1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, required=True)
    ap.add_argument("--ny", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--bc", default="dddd")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    D, vx, vy, dt = 0.05, 0.5, 0.25, 0.1
    bc = csim.bc_codes(args.bc)
    dec = csim.decomp_init(world, rank, args.nx, args.ny)
    nxl, nyl, xo, yo = dec.nx_local, dec.ny_local, dec.x_offset, dec.y_offset

    # decomposed run: this rank's tile
    u = np.zeros((nyl + 2, nxl + 2))
    u[1:-1, 1:-1] = field_rows(args.nx, args.ny, yo, nyl, xo, nxl)
    st = csim.Stepper(dec, 1.0, 1.0, bc)
    st.set_option("external_halo", 1)
    st.upload(u)
    advance(st, list(dec.nbr), D, dt, vx, vy, args.steps)
    mine = st.download()
    st.close()
    dist.barrier()

    # single-rank run of the whole grid on the same GPU, one rank at a time (each holds two whole fields)
    ok = True
    for turn in range(world):
        if turn == rank:
            whole = np.zeros((args.ny + 2, args.nx + 2))
            whole[1:-1, 1:-1] = field_rows(args.nx, args.ny, 0, args.ny, 0, args.nx)
            s1 = csim.Stepper.single(args.nx, args.ny, 1.0, 1.0, bc)
            s1.upload(whole)
            del whole
            s1.run(D, dt, vx, vy, args.steps)
            ref = s1.download()
            s1.close()
            want = ref[yo:yo + nyl + 2, xo:xo + nxl + 2]
            mask = np.zeros(mine.shape, bool)
            mask[1:-1, 1:-1] = True                       # interior
            phys = [n < 0 for n in dec.nbr]               # ghost lines of physical sides (span 1..n)
            mask[1:-1, 0] |= phys[0]
            mask[1:-1, -1] |= phys[1]
            mask[0, 1:-1] |= phys[2]
            mask[-1, 1:-1] |= phys[3]
            ok = bool(np.array_equal(mine[mask], want[mask]))
            if not ok:
                d = np.abs(mine - want) * mask
                print(f"rank {rank}: max |diff| {d.max()} at {np.unravel_index(d.argmax(), d.shape)}", flush=True)
            del ref, want
        dist.barrier()
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        print(f"FULLSIZE world={world} dims={dec.dims[0]}x{dec.dims[1]} grid={args.nx}x{args.ny} bc={args.bc} "
              f"steps={args.steps} ok={all(flags)}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()
