"""The diffusion-only flavour of the multi-step sweep (BASELINE configs[1]: vx = vy = 0).

With v == 0 the reference (src/advection.cpp:13-33 after src/diffusion.cpp:9-16) still evaluates
`o + (-dt) * (0 * dudx + 0 * dudy)`; the screened interior body leaves those seven operations out, which is the
same bits unless a value is not finite / near overflow (0 * inf = NaN) or o is -0 (-0 + +0 = +0) — tiles that load
such a value are recomputed with the reference's own sequence.  Everything here is compared BIT for bit (integer
views: +0 and -0 differ), NaN cells by position."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cpu_oracle as ora
from test_gpu_comm import CORNERLESS, self_neighbor_decomp, torus_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def csim():
    pkg = load_package()
    pkg.lib()
    assert pkg.device_count() >= 1, "no GPU visible"
    pkg.set_device(0)
    return pkg


def same_bits(got, want):
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    ok = ~np.isnan(want)
    return np.array_equal(got[ok].view(np.int64), want[ok].view(np.int64))


def nasty_field(nx, ny, seed, nonfinite=True):
    """random interior with everything the screen has to catch: blocks of +0 and of -0, single -0 cells, the smallest
    negative subnormals next to zeros (their products underflow to -0), values around the overflow screen, Inf, NaN"""
    rng = np.random.default_rng(seed)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = 0.25, -0.5, 0.75, -1.25
    u0[20:60, 130:260] = 0.0
    u0[25:55, 140:200] = -0.0
    u0[70:110, 300:420] = 0.0
    for j, i in rng.integers(0, [40, 120], size=(40, 2)):
        u0[70 + j, 300 + i] = -0.0
    for j, i in rng.integers(0, [40, 120], size=(40, 2)):
        u0[70 + j, 300 + i] = -5e-324
    u0[5:15, 450:520] = -0.0          # -0 next to ordinary values
    # zeros of both signs among the smallest subnormals: here a sweep that simply left the advection term out gives
    # other zero signs than the reference after two or more steps (skip_model below proves it for this very field)
    u0[112:152, 320:420] = rng.choice(np.array([-0.0, 0.0, -5e-324, 5e-324, -1e-323, 1e-323, -0.0, -0.0]), (40, 100))
    u0[120:130, 250:300] *= 1e-310
    if nonfinite:
        u0[30:38, 500:530] *= 1e300
        u0[90:97, 560:590] = 1.2e308 * np.sign(u0[90:97, 560:590])
        u0[140, 330] = np.inf
        u0[100, 600] = -np.inf
        u0[45, 640] = np.nan
    return u0


def skip_model(u0, k, steps):
    """what an UNSCREENED diffusion-only sweep would compute (unit spacing, Dirichlet(0) ghosts): c + k * lap and nothing else"""
    u = u0.copy()
    u[0, :] = u[-1, :] = 0.0
    u[:, 0] = u[:, -1] = 0.0
    for _ in range(steps):
        c, W, E, S, N = u[1:-1, 1:-1], u[1:-1, :-2], u[1:-1, 2:], u[:-2, 1:-1], u[2:, 1:-1]
        o = c + k * (((E - 2.0 * c) + W) + ((N - 2.0 * c) + S))
        u[1:-1, 1:-1] = o
    return u


@pytest.mark.parametrize("dx,dy", [(1.0, 1.0), (0.5, 2.0), (0.7, 1.3)])
@pytest.mark.parametrize("fuse", [2, 3, 4, 5, 6, 7, -1])
def test_zero_velocity_is_the_reference_bit_for_bit_signed_zeros_included(csim, dx, dy, fuse):
    nx, ny = 700, 160
    steps = 17 if fuse < 0 else fuse + 3
    D = 0.05
    dt = min(0.1, csim.safe_dt(dx, dy, 0.0, 0.0, D))
    for bc, vx, vy, nonfinite in [("dddd", 0.0, 0.0, True), ("npdn", 0.0, 0.0, False), ("pnnd", -0.0, 0.0, False),
                                  ("dddd", 0.0, -0.0, False)]:
        u0 = nasty_field(nx, ny, 5 + len(bc), nonfinite)
        want = u0.copy()
        with np.errstate(all="ignore"):
            ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes(bc), steps)
        if bc == "dddd" and not nonfinite and (dx, dy) == (1.0, 1.0):
            # the case has teeth: leaving the term out WITHOUT the -0 screen is visibly not the reference
            naive = skip_model(u0, dt * D, steps)
            assert not np.array_equal(naive[1:-1, 1:-1].view(np.int64), want[1:-1, 1:-1].view(np.int64))
            assert np.array_equal(naive[1:-1, 1:-1], want[1:-1, 1:-1])   # ... and only in the signs of zeros
        for on in (1, 0):
            st = csim.Stepper.single(nx, ny, dx, dy, csim.bc_codes(bc))
            for k, v in dict(fuse=fuse, rows_per_chunk=18, fused_2c=on).items():
                st.set_option(k, v)
            st.upload(u0)
            st.run(D, dt, vx, vy, steps)
            # the IEEE-division form (0.7, 1.3) keeps the full update; fused_2c = 0 switches every screened body off
            assert st.get_option("diffusion_only_active") == (1 if on and (dx, dy) != (0.7, 1.3) else 0)
            got = st.download()
            st.close()
            assert same_bits(got, want), (bc, vx, vy, on)


@pytest.mark.parametrize("dx,dy", [(1.0, 1.0), (0.5, 2.0), (0.7, 1.3)])
@pytest.mark.parametrize("fuse", [2, 4, 5, 6, 7, -1])
def test_one_zero_velocity_component_is_the_reference_bit_for_bit(csim, dx, dy, fuse):
    """vx = 0 or vy = 0 alone (the reference's own configs/dev.yaml has vy = 0): the screened interior body leaves the
    three operations of the zero component out — same screen, same argument, both upwind directions of the live one"""
    nx, ny = 700, 160
    steps = 17 if fuse < 0 else fuse + 3
    D = 0.05
    for k, (bc, vx, vy, nonfinite) in enumerate([("dddd", 0.5, 0.0, True), ("npdn", -0.5, 0.0, False), ("pnnd", 0.0, 0.25, False),
                                                 ("dnpd", 0.0, -0.25, False), ("dddd", 0.5, -0.0, False), ("nnnn", -0.0, -0.25, False)]):
        dt = min(0.1, csim.safe_dt(dx, dy, vx, vy, D))
        u0 = nasty_field(nx, ny, 40 + k, nonfinite)
        want = u0.copy()
        with np.errstate(all="ignore"):
            ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes(bc), steps)
        for on in (1, 0):
            st = csim.Stepper.single(nx, ny, dx, dy, csim.bc_codes(bc))
            for key, v in dict(fuse=fuse, rows_per_chunk=18, fused_2c=on).items():
                st.set_option(key, v)
            st.upload(u0)
            st.run(D, dt, vx, vy, steps)
            assert st.get_option("diffusion_only_active") == 0
            got = st.download()
            st.close()
            assert same_bits(got, want), (bc, vx, vy, on)


def test_flavour_follows_the_velocity_from_run_to_run(csim):
    """one stepper, runs with v = 0 and v != 0 in turn: each run uses its own flavour (and re-tunes), results as the oracle's"""
    nx, ny = 1200, 900
    D, dt = 0.05, 0.1
    u0 = nasty_field(nx, ny, 3, nonfinite=False)
    want = u0.copy()
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("dnpd"))
    st.upload(u0)
    for vx, vy, steps in [(0.0, 0.0, 30), (0.5, -0.25, 23), (0.0, 0.0, 9), (0.0, 0.25, 8), (-0.5, 0.0, 15), (0.5, 0.25, 6)]:
        ora.run_single(want, 1.0, 1.0, D, vx, vy, dt, ora.bc_codes("dnpd"), steps)
        st.run(D, dt, vx, vy, steps)
        assert st.get_option("diffusion_only_active") == (1 if vx == 0.0 and vy == 0.0 else 0)
        assert same_bits(st.download(), want), (vx, vy, steps)
    st.close()


@pytest.mark.parametrize("overlap", [0, 1, 3, 5])
def test_zero_velocity_across_ranks(csim, overlap):
    """the exchange path (self-linked torus, deep faces in 8 directions) under the diffusion-only flavour"""
    nx, ny, steps = 1024, 300, 23
    D, dt = 0.1, 0.1
    u0 = nasty_field(nx, ny, 11, nonfinite=False)
    u0[0, :] = u0[-1, :] = 0.0
    u0[:, 0] = u0[:, -1] = 0.0
    want = torus_oracle(u0, 1.0, 1.0, D, 0.0, 0.0, dt, steps)
    st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, (1, 1, 1, 1)), 1.0, 1.0, csim.bc_codes("dddd"))
    st.comm_init(csim.comm_unique_id())
    st.set_option("overlap", overlap)
    st.upload(u0)
    st.run(D, dt, 0.0, 0.0, 9)
    st.run(D, dt, 0.0, 0.0, steps - 9)
    assert st.get_option("diffusion_only_active") == 1
    got = st.download()
    st.close()
    m = CORNERLESS(ny, nx)
    assert np.array_equal(got[m].view(np.int64), want[m].view(np.int64))


def test_config2_4096_diffusion_periodic_full_field(csim):
    """BASELINE configs[1] as specified (4096^2, D = 1, v = 0, dt = 0.1, all Periodic), whole field against the oracle"""
    n, steps = 4096, 21
    u0 = ora.gaussian_global(n, n)
    want = u0.copy()
    ora.run_single(want, 1.0, 1.0, 1.0, 0.0, 0.0, 0.1, ora.bc_codes("pppp"), steps)
    st = csim.Stepper.single(n, n, 1.0, 1.0, csim.bc_codes("pppp"))
    st.upload(u0)
    st.run(1.0, 0.1, 0.0, 0.0, steps)
    assert st.get_option("diffusion_only_active") == 1
    got = st.download()
    st.close()
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
