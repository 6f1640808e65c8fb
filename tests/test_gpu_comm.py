"""The RCCL halo-exchange path on ONE GPU.  A 1-rank communicator whose four neighbours are
the rank itself turns the tile into a true torus, which drives every piece of the multi-GPU
step — edge-pack kernel, grouped ncclSend/ncclRecv on the second stream, event hand-off, unpack
in the ghost fill, overlap on/off — through real RCCL calls.  (The reference itself never
wraps; the torus is only the cheapest way to exercise the exchange on one device.
Multi-rank decomposition logic is covered on CPU by tests/test_multirank_gloo.py.)"""
import os

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cpu_oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def csim():
    pkg = load_package()
    pkg.lib()
    pkg.set_device(0)
    return pkg


def torus_oracle(u0, dx, dy, D, vx, vy, dt, steps, sides=(1, 1, 1, 1), bc=(0, 0, 0, 0)):
    """reference step with the ghosts of the `sides` flagged 1 wrapped from the opposite edge
    (what exchange_halos delivers when the neighbour is the rank itself)."""
    u = u0.copy()
    tmp = u0.copy()
    phys = [0 if s else 1 for s in sides]
    for _ in range(steps):
        if sides[0]:
            u[1:-1, 0] = u[1:-1, -2]
        if sides[1]:
            u[1:-1, -1] = u[1:-1, 1]
        if sides[2]:
            u[0, 1:-1] = u[-2, 1:-1]
        if sides[3]:
            u[-1, 1:-1] = u[1, 1:-1]
        ora.step_tile(u, tmp, dx, dy, D, vx, vy, dt, bc, phys)
        u, tmp = tmp, u
    return u


def self_neighbor_decomp(csim, nx, ny, sides):
    d = csim.decomp_init(1, 0, nx, ny)
    for k in range(4):
        d.nbr[k] = 0 if sides[k] else csim.NO_NEIGHBOR
    return d


# neighbour links come in opposite pairs (if A's left is B, B's right is A), so the self-linked
# sides are {left,right} and/or {bottom,top}
@pytest.mark.parametrize("sides,bc", [((1, 1, 1, 1), "dddd"), ((1, 1, 0, 0), "ddnd"),
                                      ((0, 0, 1, 1), "npdd"), ((1, 1, 0, 0), "ddpp")])
@pytest.mark.parametrize("overlap", [1, 0, 3, 4, 5])
@pytest.mark.parametrize("shape", [(300, 170, 7), (256, 170, 9), (1024, 300, 12), (128, 2, 8)])
def test_self_exchange_torus(csim, sides, bc, overlap, shape):
    # fused passes across "ranks": deep faces and corner blocks in 8 directions, frame tiles first,
    # exchange overlapped with the remaining tiles
    nx, ny, steps = shape
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(17)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    codes = csim.bc_codes(bc)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)

    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", overlap)
    st.upload(u0)
    st.run(D, dt, vx, vy, 3)
    st.run(D, dt, vx, vy, steps - 3)
    got = st.download()
    # the single-step path must give the very same field
    st.set_option("fuse", 0)
    st.upload(u0)
    st.run(D, dt, vx, vy, steps)
    got1 = st.download()
    st.close()
    assert np.array_equal(got[1:-1, 1:-1], got1[1:-1, 1:-1])
    # corners are never exchanged (reference leaves them undefined)
    mask = np.ones(got.shape, bool)
    mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    assert np.array_equal(got[mask], want[mask]), float(np.abs(got - want)[mask].max())


def test_torus_at_tile_scale_every_schedule_equals_the_oracle(csim):
    """a per-GPU-tile-sized torus (many strips x many chunks, so the frame / bulk split and the
    8-direction deep faces are all in play): every exchange schedule — serial single steps, the
    overlapped 6-step passes, merged and bulk-first launches — must reproduce the ORACLE's torus bit for bit."""
    nx, ny, steps = 2048, 4096, 44
    D, vx, vy, dt = 0.1, -0.5, 0.25, 0.1
    d = self_neighbor_decomp(csim, nx, ny, (1, 1, 1, 1))
    # hotspot sitting on a torus corner: all 8 faces carry data
    u0 = ora.gaussian_global(nx, ny, sigma_frac=0.02, xc_frac=0.03, yc_frac=0.97)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps)[1:-1, 1:-1]
    assert want[0, 0] > 0 and want[-1, -1] > 0 and want[0, -1] > 0 and want[-1, 0] > 0
    for opts in [dict(overlap=0, fuse=0), dict(overlap=1, fuse=-1), dict(overlap=0, fuse=-1), dict(overlap=1, fuse=6), dict(overlap=3, fuse=6),
                 dict(overlap=1, fuse=4, rows_per_chunk=64), dict(overlap=1, fuse=3), dict(overlap=1, fuse=5),
                 dict(overlap=3, fuse=-1), dict(overlap=3, fuse=5, rows_per_chunk=40), dict(overlap=3, fuse=2),
                 dict(overlap=4, fuse=-1), dict(overlap=4, fuse=7), dict(overlap=5, fuse=-1), dict(overlap=4, fuse=3)]:
        st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes("dddd"))
        st.comm_init(csim.comm_unique_id())
        for k, v in opts.items():
            st.set_option(k, v)
        st.upload(u0)
        st.run(D, dt, vx, vy, steps)
        out = st.download_interior()
        st.close()
        assert np.array_equal(out, want), opts


@pytest.mark.parametrize("sides,bc", [((1, 1, 0, 0), "ddnp"), ((0, 0, 1, 1), "pndd"), ((1, 1, 0, 0), "ddnn")])
@pytest.mark.parametrize("overlap", [1, 0, 3, 4, 5])
def test_torus_mixed_physical_and_linked_sides_depth6(csim, sides, bc, overlap):
    """linked sides next to physical Neumann / Periodic sides on a tile tall and wide enough for the
    frame / bulk split, three 6-step passes (overlapped exchange, comm-stream pre-unpack + ghost fill
    with its halo extension, closing FinLines ghost fill): the FULL array incl. the ghost ring must be
    the oracle's."""
    nx, ny, steps = 1160, 300, 18
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(23)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = 0.5, -0.25, 0.125, -1.5   # Periodic ghosts must survive
    codes = csim.bc_codes(bc)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", overlap)
    st.set_option("fuse", 6)          # (auto would take depth 4 on a tile this small)
    st.set_option("profile", 1)
    st.upload(u0)
    st.run(D, dt, vx, vy, steps)
    got = st.download()
    assert st.kernel_time(6)[1] == 3
    st.close()
    mask = np.ones(got.shape, bool)
    mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False   # corners are never exchanged
    assert np.array_equal(got[mask], want[mask]), float(np.abs(got - want)[mask].max())


@pytest.mark.parametrize("sides,bc", [((1, 1, 0, 0), "ddnp"), ((0, 0, 1, 1), "pndd"), ((1, 1, 0, 0), "ddnn"), ((0, 0, 1, 1), "nndd"),
                                      ((1, 1, 1, 1), "dddd")])
@pytest.mark.parametrize("overlap", [3, 4, 1])
def test_mixed_sides_depths_7_6_5_on_one_stepper(csim, sides, bc, overlap):
    """the comm-stream chain (faces -> halo cells, ghost ring, extension of the physical edges over the halo,
    corners) with linked sides next to physical Neumann / Periodic / Dirichlet sides, passes of depth 7, 6 and 5 on
    the same stepper: the full array incl. the ghost ring against the oracle's torus."""
    nx, ny, steps = 1160, 300, 18
    D, vx, vy, dt = 0.05, -0.5, 0.25, 0.1
    rng = np.random.default_rng(29)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = 0.5, -0.25, 0.125, -1.5
    codes = csim.bc_codes(bc)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", overlap)
    st.upload(u0)
    for depth, n in ((7, 7), (6, 6), (5, 5)):
        st.set_option("fuse", depth)
        st.run(D, dt, vx, vy, n)
    got = st.download()
    st.close()
    mask = np.ones(got.shape, bool)
    mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    assert np.array_equal(got[mask], want[mask]), float(np.abs(got - want)[mask].max())


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("shape", [(1161, 301, 6), (897, 130, 7), (300, 171, 5), (2049, 64, 4), (127, 40, 3), (64, 515, 2)])
@pytest.mark.parametrize("sides,bc", [((1, 1, 1, 1), "dddd"), ((1, 1, 0, 0), "ddpn"), ((0, 0, 1, 1), "pndd")])
def test_merged_launch_direct_faces_vs_pack_kernel(csim, direct, shape, sides, bc):
    """schedule 3 with the frame wavefronts writing the NEXT pass's faces straight into the RCCL send buffers
    ("direct_faces" = 1, default) and with the pack kernel on the comm stream (0): odd and even widths (the lane
    that owns column nx carries the right ghost entry), tiles of one strip and of many, every depth, Periodic
    ghosts (which travel inside the faces) beside linked sides — the full array against the oracle's torus."""
    nx, ny, depth = shape
    steps = 3 * depth + 2
    D, vx, vy, dt = 0.05, -0.5, 0.25, 0.1
    rng = np.random.default_rng(nx + ny)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = 0.5, -0.25, 0.125, -1.5
    codes = csim.bc_codes(bc)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", 3)
    st.set_option("fuse", depth)
    st.set_option("direct_faces", direct)
    assert st.get_option("direct_faces") == direct
    st.upload(u0)
    st.run(D, dt, vx, vy, steps)
    got = st.download()
    st.close()
    mask = np.ones(got.shape, bool)
    mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    assert np.array_equal(got[mask], want[mask]), float(np.abs(got - want)[mask].max())


def test_torus_on_an_8192_square_tile_bulk_with_tail_region(csim):
    """the 4-GPU tile: its bulk is more than two rounds of wavefronts, so the merged and the bulk-first launches
    carry a TAIL region of half-height chunks behind the XCD-remapped main tiles (Tiling::tail_blocks) — the
    block -> tile map of a three-part grid (frame, main, tail).  Every schedule must leave the field the serial
    single-step schedule leaves (which the smaller torus tests tie to the oracle), bit for bit."""
    n, steps = 8192, 19
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    d = self_neighbor_decomp(csim, n, n, (1, 1, 1, 1))
    ref = None
    for opts in [dict(overlap=0, fuse=0), dict(overlap=3, fuse=6), dict(overlap=4, fuse=7), dict(overlap=5, fuse=-1),
                 dict(overlap=3, fuse=6, tail_split=0), dict(overlap=1, fuse=5, rows_per_chunk=50),
                 dict(overlap=3, fuse=6, direct_faces=0)]:
        st = csim.Stepper(d, 1.0, 1.0, csim.bc_codes("dddd"))
        st.comm_init(csim.comm_unique_id())
        for k, v in opts.items():
            st.set_option(k, v)
        st.init_gaussian(1.0, 0.02, 0.97, 0.03)   # hotspot on a torus corner: all 8 faces carry data
        mass0 = st.sum()
        st.run(D, dt, vx, vy, steps)
        out = st.download_interior()
        mass1 = st.sum()
        st.close()
        assert abs(mass1 - mass0) <= 1e-12 * abs(mass0)   # a torus loses nothing
        if ref is None:
            ref = out
            assert ref[0, 0] > 0 and ref[-1, -1] > 0 and ref[0, -1] > 0 and ref[-1, 0] > 0
        else:
            assert np.array_equal(out, ref), opts


def test_fuzz_random_torus_cases_vs_oracle(csim):
    """80 seeded random cases of the RCCL path on the self-linked torus: tile shape, which side pairs are linked,
    the BC of the physical sides, physics, step count cut into two run() calls, pass depth, exchange schedule —
    the full array (ghost ring included, corners excepted: SURVEY Q7) against the oracle's torus."""
    rng = np.random.default_rng(int(os.environ.get("CSIM_FUZZ_SEED", "777")))   # (other seeds: extended soak runs)
    for case in range(80):
        nx, ny = int(rng.integers(8, 700)), int(rng.integers(8, 260))
        sides = [(1, 1, 1, 1), (1, 1, 0, 0), (0, 0, 1, 1)][int(rng.integers(0, 3))]
        bc = "".join(rng.choice(list("dnp"), 4))
        D = float(rng.choice([0.0, 0.05, 0.2]))
        vx, vy = float(rng.choice([0.5, -0.5, 0.0])), float(rng.choice([0.25, -0.25, 0.0]))
        dt = 0.1 if (D or vx or vy) else 0.1
        steps = int(rng.integers(2, 26))
        opts = dict(overlap=int(rng.choice([0, 1, 3, 4, 5])), fuse=int(rng.choice([-1, -1, 0, 2, 3, 4, 5, 6, 7])),
                    rows_per_chunk=int(rng.choice([0, 0, 3, 20])), direct_faces=case % 2)
        u0 = np.zeros((ny + 2, nx + 2))
        u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
        u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = rng.standard_normal(4)
        codes = csim.bc_codes(bc)
        want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)
        st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
        st.comm_init(csim.comm_unique_id())
        for k, v in opts.items():
            st.set_option(k, v)
        st.upload(u0)
        a = int(rng.integers(1, steps))
        st.run(D, dt, vx, vy, a)
        st.run(D, dt, vx, vy, steps - a)
        got = st.download()
        st.close()
        mask = np.ones(got.shape, bool)
        mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
        assert np.array_equal(got[mask], want[mask]), (case, nx, ny, sides, bc, D, vx, vy, steps, a, opts,
                                                       float(np.abs(got - want)[mask].max()))


@pytest.mark.parametrize("overlap", [3, 4, 1])
def test_keep_warm_leaves_the_run_untouched(csim, overlap):
    """csim_stepper_keep_warm (bench.py's wait between the cross-rank barrier and the timed region): whole-tile
    launches into the scratch buffer between run() calls — also between a run that left faces pre-unpacked for
    nobody and the next one — must not change a bit of what the runs produce, ghost ring included."""
    import time
    nx, ny = 1160, 300
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    sides, bc = (1, 1, 0, 0), "ddnp"
    rng = np.random.default_rng(41)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :], u0[-1, :] = 0.5, -0.25
    codes = csim.bc_codes(bc)
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, 6 + 13 + 1, sides, codes)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", overlap)
    st.set_option("fuse", 6)
    st.upload(u0)
    t0 = time.perf_counter()
    st.keep_warm(D, dt, vx, vy, 0.02)
    took = time.perf_counter() - t0
    assert took < 0.05          # returns by the deadline (plus call overhead), GPU idle
    st.run(D, dt, vx, vy, 6)
    st.keep_warm(D, dt, vx, vy, 0.005)
    st.run(D, dt, vx, vy, 13)
    st.keep_warm(D, dt, vx, vy, 0.0)
    st.run(D, dt, vx, vy, 1)
    got = st.download()
    st.close()
    mask = np.ones(got.shape, bool)
    mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    assert np.array_equal(got[mask], want[mask])


def test_exchange_halos_alone(csim):
    """reference tests/simulation/unit/test_halo.cpp:36-56 restated: after exchange_halos every
    ghost face on a neighbour side holds the neighbour's edge cells, physical sides untouched."""
    nx, ny = 40, 24
    rng = np.random.default_rng(4)
    u0 = np.full((ny + 2, nx + 2), -1.0)
    u0[1:-1, 1:-1] = rng.random((ny, nx))
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, (1, 1, 0, 0)), 1.0, 1.0, csim.bc_codes("dddd"))
    st.comm_init(csim.comm_unique_id())
    st.upload(u0)
    st.exchange_halos()
    g = st.download()
    st.close()
    assert np.array_equal(g[1:-1, 0], u0[1:-1, -2]) and np.array_equal(g[1:-1, -1], u0[1:-1, 1])
    assert (g[0, :] == -1.0).all() and (g[-1, :] == -1.0).all()  # no y-neighbours: untouched
    assert np.array_equal(g[1:-1, 1:-1], u0[1:-1, 1:-1])


def test_multi_rank_run_needs_comm(csim):
    st = csim.Stepper(self_neighbor_decomp(csim, 16, 16, (1, 0, 0, 0)), 1.0, 1.0, [0, 0, 0, 0])
    with pytest.raises(csim.CsimError):
        st.run(0.1, 0.1, 0.0, 0.0, 1)
    st.close()


def CORNERLESS(ny, nx):
    """corner ghosts are never exchanged at depth 1 (the reference leaves them undefined)"""
    m = np.ones((ny + 2, nx + 2), bool)
    m[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    return m


def test_checksum_is_decomposition_invariant_and_matches_numpy(csim):
    """csim_stepper_checksum: position-weighted sum of the interior's bit patterns; the per-tile values of any
    decomposition add up (mod 2^64) to the checksum of the whole field — here against the numpy restatement."""
    nx, ny = 301, 173
    rng = np.random.default_rng(3)
    g = rng.standard_normal((ny, nx))
    g[5, 7], g[6, 7] = 1e-310, -0.0          # a subnormal and a negative zero: bit patterns, not values
    want = csim.checksum_host(g)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = g
    st = csim.Stepper.single(nx, ny)
    st.upload(u0)
    assert st.checksum() == want
    g2 = g.copy()
    g2[100, 200], g2[100, 201] = g[100, 201], g[100, 200]   # two cells swapped: same multiset of values
    u0[1:-1, 1:-1] = g2
    st.upload(u0)
    assert st.checksum() != want and st.checksum() == csim.checksum_host(g2)
    st.close()
    for world in (2, 6, 8):
        total = 0
        for r in range(world):
            dec = csim.decomp_init(world, r, nx, ny)
            t = np.zeros((dec.ny_local + 2, dec.nx_local + 2))
            t[1:-1, 1:-1] = g[dec.y_offset:dec.y_offset + dec.ny_local, dec.x_offset:dec.x_offset + dec.nx_local]
            s = csim.Stepper(dec)
            s.upload(t)
            part = s.checksum()
            assert part == csim.checksum_host(t[1:-1, 1:-1], dec.x_offset, dec.y_offset, nx)
            total = (total + part) % (1 << 64)
            s.close()
        assert total == want, world


def test_sync_timeout_and_injected_stall_are_reported_not_hung(csim):
    """A comm stream that never drains (option "test_stall": parked on a signal value nobody publishes — what a
    lost flag or a dead peer looks like from the host) must come back from csim_stepper_sync as CSIM_ERR_TIMEOUT
    when "sync_timeout_ms" is set, leave the stepper usable after the release, and not change any result."""
    nx, ny, steps = 512, 256, 14
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(23)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, (1, 1, 1, 1)), 1.0, 1.0, csim.bc_codes("dddd"))
    st.comm_init(csim.comm_unique_id())
    st.upload(u0)
    st.run(D, dt, vx, vy, 7)
    st.sync()
    st.set_option("sync_timeout_ms", 300)
    st.set_option("test_stall", 1)
    with pytest.raises(csim.CsimError) as e:
        st.sync()
    assert e.value.code == 6 and "sync_timeout_ms" in str(e.value)
    st.set_option("test_stall", 0)
    st.sync()
    st.set_option("sync_timeout_ms", 0)
    st.run(D, dt, vx, vy, steps - 7)
    assert np.array_equal(st.download()[CORNERLESS(ny, nx)], want[CORNERLESS(ny, nx)])
    st.close()


def test_borrowed_communicator_runs_a_second_stepper(csim):
    """csim_stepper_comm_share: a small parity case beside the production tile on ONE communicator (what
    bench.py's preflight does at N > 1), both bit-identical to the oracle, the borrower closed first."""
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(29)
    big = csim.Stepper(self_neighbor_decomp(csim, 1024, 512, (1, 1, 1, 1)), 1.0, 1.0, csim.bc_codes("dddd"))
    big.comm_init(csim.comm_unique_id())
    small = csim.Stepper(self_neighbor_decomp(csim, 96, 40, (1, 1, 1, 1)), 1.0, 1.0, csim.bc_codes("dddd"))
    small.comm_share(big)
    with pytest.raises(csim.CsimError):
        small.comm_share(big)
    fields = {}
    for st, (nx, ny) in ((big, (1024, 512)), (small, (96, 40))):
        u0 = np.zeros((ny + 2, nx + 2))
        u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
        st.upload(u0)
        fields[st] = u0
    for overlap in (0, 4, 3):
        for st in (small, big):
            st.set_option("overlap", overlap)
            st.upload(fields[st])
            st.run(D, dt, vx, vy, 13)
            st.sync()   # one stepper's exchanges at a time on a shared communicator
            m = CORNERLESS(st.ny, st.nx)
            assert np.array_equal(st.download()[m], torus_oracle(fields[st], 1.0, 1.0, D, vx, vy, dt, 13)[m]), overlap
    small.close()
    big.run(D, dt, vx, vy, 2)
    big.sync()
    big.close()


@pytest.mark.parametrize("bc", ["ddnd", "dddn", "dddd", "ddpp", "ddpn", "dddp"])
def test_physical_bottom_and_top_next_to_linked_sides_every_depth_and_schedule(csim, bc):
    """A tile like the mid-x tiles of the 4 x 2 grid: left / right neighbours, PHYSICAL bottom / top.  Its frame holds
    thin generic bands along Dirichlet / Periodic sides (T-1 rows: the tiles above start at row T and read the ghost
    line as level-0 input — in a bulk-first pass before that pass's own ghost fill, hence the one-off fill of the
    physical ghost lines at the start of a run) and full-height bands along Neumann sides.  Uploaded ghost lines are
    random: Periodic sides must keep them, Dirichlet / Neumann sides must overwrite them before anything reads them."""
    nx, ny = 1024, 300
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(31)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :] = rng.standard_normal(nx + 2)
    u0[-1, :] = rng.standard_normal(nx + 2)
    sides, codes = (1, 1, 0, 0), csim.bc_codes(bc)
    mask = CORNERLESS(ny, nx)
    for steps, fuse in ((7, 7), (5, 5), (6, 6), (4, 4), (20, -1)):
        want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps, sides, codes)
        for overlap in (4, 3, 1):
            st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, sides), 1.0, 1.0, codes)
            st.comm_init(csim.comm_unique_id())
            st.set_option("overlap", overlap)
            st.set_option("fuse", fuse)
            st.upload(u0)
            st.run(D, dt, vx, vy, steps)
            got = st.download()
            st.close()
            assert np.array_equal(got[mask], want[mask]), (bc, steps, fuse, overlap, int((got[mask] != want[mask]).sum()))
