"""Host-staged halo transport for the stepper's `external_halo` mode over torch.distributed
point-to-point messages (gloo): what a caller's own MPI would do with
csim_stepper_halo_pack/_unpack and csim_stepper_faces_pack/_unpack (reference src/halo.cpp:28-46
posts the same Irecv/Isend pairs).  Used by the multi-rank tests (several ranks may share one GPU;
RCCL refuses that) and by bench.py as the fall-back when the RCCL communicator cannot be built."""
import numpy as np
import torch
import torch.distributed as dist

OPPOSITE = {0: 1, 1: 0, 2: 3, 3: 2}


def opposite8(d):
    return d ^ 1 if d < 4 else 11 - d


def exchange(lines, nbr):
    """send my edge line of side k to nbr[k]; receive the neighbour's (its opposite side)."""
    reqs, got = [], [None] * 4
    for k in range(4):
        if nbr[k] >= 0:
            got[k] = torch.empty(lines[k].shape[0], dtype=torch.float64)
            reqs.append(dist.irecv(got[k], src=nbr[k], tag=OPPOSITE[k]))
    for k in range(4):
        if nbr[k] >= 0:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(lines[k])), dst=nbr[k], tag=k))
    for r in reqs:
        r.wait()
    return [g.numpy() if g is not None else None for g in got]


def exchange8(faces, peers):
    """deep faces, 8 directions (L R B T BL BR TL TR); None where there is no peer."""
    reqs, got = [], [None] * 8
    for d in range(8):
        if peers[d] >= 0:
            got[d] = torch.empty(faces[d].shape[0], dtype=torch.float64)
            reqs.append(dist.irecv(got[d], src=peers[d], tag=opposite8(d)))
    for d in range(8):
        if peers[d] >= 0:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(faces[d])), dst=peers[d], tag=d))
    for r in reqs:
        r.wait()
    return [g.numpy() if g is not None else None for g in got]


def advance(st, nbr, D, dt, vx, vy, nsteps):
    """nsteps reference steps of a stepper in external_halo mode: fused passes with deep faces while
    at least three steps remain, then single steps with 1-cell faces (the schedule every rank
    derives identically from csim_stepper_fuse_limit)."""
    depth = st.fuse_limit()
    remaining = nsteps
    while remaining >= 3 and depth >= 2:
        t = min(depth, remaining - 1)
        peers, _ = st.faces_neighbors(t)
        st.faces_unpack(t, exchange8(st.faces_pack(t), peers))
        st.run(D, dt, vx, vy, t)
        remaining -= t
    while remaining > 0:
        st.halo_unpack(exchange(st.halo_pack(), nbr))
        st.run(D, dt, vx, vy, 1)
        remaining -= 1
