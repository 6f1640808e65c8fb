"""The message order of the multi-GPU halo exchange, checked without any GPU.

csim_exchange_plan (pure host arithmetic in libcsim.so) is the ONE place that decides which faces a
rank sends and receives inside its ncclGroupStart/End and in which order; the stepper's post_exchange /
post_exchange2 walk exactly that plan.  RCCL matches the messages of a pair of ranks in posting order,
so the first real 2/4/8-GPU run deadlocks or scrambles faces unless, for EVERY ordered pair (a, b):
the k-th send a posts to b has the length of — and the direction opposite to — the k-th receive b
posts for a, and nobody waits for a message that is never sent.  Checked here for the process grids
MPI_Dims_create yields for 2, 4, 6, 8, 3 and 12 ranks, every depth 1..7, with and without remainder
tiles (reference decomposition: src/decomp.cpp:24-33; reference exchange: src/halo.cpp:28-46)."""
import collections

import pytest

from __graft_entry__ import load_package

csim = load_package()

OPPOSITE = {0: 1, 1: 0, 2: 3, 3: 2, 4: 7, 7: 4, 5: 6, 6: 5}   # L R B T BL BR TL TR
OFFSET = {0: (-1, 0), 1: (1, 0), 2: (0, -1), 3: (0, 1), 4: (-1, -1), 5: (1, -1), 6: (-1, 1), 7: (1, 1)}


def plans(world, nx, ny, depth):
    decs = [csim.decomp_init(world, r, nx, ny) for r in range(world)]
    return decs, [csim.exchange_plan(d, depth) for d in decs]


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("world,dims", [(2, (2, 1)), (4, (2, 2)), (6, (3, 2)), (8, (4, 2)), (3, (3, 1)), (12, (4, 3))])
@pytest.mark.parametrize("grid", [(16384, 16384), (1000, 777), (515, 67), (97, 61)])
def test_every_pair_matches_in_posting_order(world, dims, grid, depth):
    nx, ny = grid
    decs, pl = plans(world, nx, ny, depth)
    assert tuple(decs[0].dims) == dims
    if depth > min(nx // dims[0], ny // dims[1]):
        pytest.skip("tile smaller than the face depth (the stepper caps the depth: fuse_cap)")
    sent = collections.defaultdict(list)      # (a, b) -> [(dir, count)] in a's posting order
    expected = collections.defaultdict(list)  # (a, b) -> [(dir, count)] b expects from a, in b's posting order
    for r, (sends, recvs) in enumerate(pl):
        for peer, d, n in sends:
            assert 0 <= peer < world and peer != r
            sent[(r, peer)].append((d, n))
        for peer, d, n in recvs:
            assert 0 <= peer < world and peer != r
            expected[(peer, r)].append((d, n))
    assert set(sent) == set(expected), "a rank waits for a peer that never sends (or the reverse)"
    for pair, msgs in sent.items():
        want = expected[pair]
        assert len(msgs) == len(want), (pair, msgs, want)
        for (ds, ns), (dr, nr) in zip(msgs, want):
            assert ns == nr and ns > 0, (pair, msgs, want)      # k-th send length == k-th receive length
            assert dr == OPPOSITE[ds], (pair, msgs, want)       # it arrives from the opposite direction


@pytest.mark.parametrize("depth", [1, 3, 6, 7])
@pytest.mark.parametrize("world", [2, 4, 6, 8])
def test_peers_are_the_geometric_neighbours_and_lengths_fit_the_tiles(world, depth):
    nx, ny = 1030, 517      # remainders on the last column / row of tiles
    decs, pl = plans(world, nx, ny, depth)
    by_coords = {(d.coords[0], d.coords[1]): r for r, d in enumerate(decs)}
    for r, (sends, recvs) in enumerate(pl):
        d = decs[r]
        dirs = [m[1] for m in sends]
        assert dirs == sorted(dirs) and len(set(dirs)) == len(dirs)
        assert sorted(m[1] for m in recvs) == dirs           # one receive per direction a face leaves in
        for peer, k, n in sends:
            ox, oy = OFFSET[k]
            assert by_coords[(d.coords[0] + ox, d.coords[1] + oy)] == peer
            if depth == 1:
                assert k < 4 and n == (d.ny_local if k < 2 else d.nx_local)      # src/halo.cpp:12-18 spans
            else:
                assert n == (depth * (d.ny_local + 2) if k < 2 else depth * (d.nx_local + 2) if k < 4 else depth * depth)
        # a direction has a peer exactly when the tile is not on that edge of the process grid
        for k, (ox, oy) in OFFSET.items():
            inside = (d.coords[0] + ox, d.coords[1] + oy) in by_coords
            assert (k in dirs) == (inside and (depth > 1 or k < 4))


def test_self_linked_torus_plan_is_consistent_too():
    """the one-GPU test topology: one rank that is its own neighbour in all eight directions"""
    d = csim.decomp_init(1, 0, 4096, 8192)
    for k in range(4):
        d.nbr[k] = 0
    for depth in range(1, 8):
        sends, recvs = csim.exchange_plan(d, depth)
        assert len(sends) == len(recvs) == (4 if depth == 1 else 8)
        for (ps, ds, ns), (pr, dr, nr) in zip(sends, recvs):   # same peer for all: pure posting order
            assert ps == pr == 0 and ns == nr and dr == OPPOSITE[ds]


def test_plan_rejects_bad_arguments():
    d = csim.decomp_init(4, 1, 64, 64)
    with pytest.raises(csim.CsimError):
        csim.exchange_plan(d, 0)
    with pytest.raises(csim.CsimError):
        csim.exchange_plan(d, 8)
