"""Parity tests proper: the HIP path, called through the C ABI (include/csim.h), against
(1) the golden vectors generated from the compiled reference and (2) the oracle on seeded
inputs.  Everything is fp64 in the reference's association order with FMA contraction off,
so the bar is BIT-EXACT (np.array_equal), far inside the north-star tolerance L_inf < 1e-10.
At BASELINE.json's full sizes, parity is checked through the stencil's domain of dependence:
after k steps a window depends only on the initial data within k cells of it, so random
windows (incl. strip / chunk seams and physical edges) are re-computed by the oracle."""
import glob
import json
import os

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cpu_oracle as ora

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RUN_FILES = sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz")))
VARIANT_CFGS = [dict(variant=1, prefetch=1), dict(variant=1, prefetch=2), dict(variant=1, prefetch=4),
                dict(variant=1, prefetch=8, rows_per_chunk=7), dict(variant=1, rows_per_chunk=1),
                dict(variant=1, xcd_swizzle=0), dict(variant=2), dict(variant=2, rows_per_chunk=5),
                dict(variant=3), dict(fuse=0), dict(fuse=2), dict(fuse=3), dict(fuse=4),
                dict(fuse=4, rows_per_chunk=3), dict(fuse=3, rows_per_chunk=1),
                dict(fuse=5), dict(fuse=6), dict(fuse=6, rows_per_chunk=2), dict(fuse=5, rows_per_chunk=1),
                dict(fuse=7), dict(fuse=7, rows_per_chunk=3)]


@pytest.fixture(scope="module")
def csim():
    pkg = load_package()
    pkg.lib()
    assert pkg.device_count() >= 1, "no GPU visible"
    pkg.set_device(0)
    assert pkg.device_name().startswith("gfx950"), pkg.device_name()
    return pkg


def _load(path):
    z = np.load(path, allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def with_ghosts(interior):
    ny, nx = interior.shape
    f = np.zeros((ny + 2, nx + 2))
    f[1:-1, 1:-1] = interior
    return f


def run_gpu(csim, u0, dx, dy, D, vx, vy, dt, bc, steps, opts=None, split=None):
    ny, nx = u0.shape[0] - 2, u0.shape[1] - 2
    st = csim.Stepper.single(nx, ny, dx, dy, bc)
    for k, v in (opts or {}).items():
        st.set_option(k, v)
    st.upload(u0)
    if split:  # several run() calls must compose exactly like one
        done = 0
        for n in split:
            st.run(D, dt, vx, vy, n)
            done += n
        assert done == steps
    else:
        st.run(D, dt, vx, vy, steps)
    out = st.download()
    st.close()
    return out


@pytest.mark.parametrize("path", RUN_FILES, ids=[os.path.basename(p)[:-4] for p in RUN_FILES])
def test_golden_runs_bit_exact_all_variants(csim, path):
    z, m = _load(path)
    dt = float(z["dt_effective"])
    assert dt == min(m["dt"], csim.safe_dt(m["dx"], m["dy"], m["vx"], m["vy"], m["D"]))
    u0 = with_ghosts(z["u0"])
    want = z["local_np1_rank0"]  # full local array of the reference's 1-rank run
    for opts in VARIANT_CFGS:
        got = run_gpu(csim, u0, m["dx"], m["dy"], m["D"], m["vx"], m["vy"], dt,
                      csim.bc_codes(m["bc"]), m["steps"], opts)
        assert np.array_equal(got, want), (os.path.basename(path), opts,
                                           float(np.abs(got - want).max()))


def test_golden_unit_steps(csim):
    z, cases = _load(os.path.join(GOLDEN, "unit_steps.npz"))
    for c in cases:
        k = c["idx"]
        u = csim.Field(c["nx"], c["ny"], 1, c["dx"], c["dy"]).upload(z[f"c{k}_u"])
        o = csim.Field(c["nx"], c["ny"], 1, c["dx"], c["dy"]).upload(z[f"c{k}_o"])
        if c["op"] == "diffusion":
            csim.diffusion_step(u, o, c["D"], c["dt"])
        else:
            csim.advection_step(u, o, c["vx"], c["vy"], c["dt"])
        assert np.array_equal(o.download(), z[f"c{k}_out"]), c
        assert np.array_equal(u.download(), z[f"c{k}_u"])


def test_golden_boundary(csim):
    z, cases = _load(os.path.join(GOLDEN, "boundary.npz"))
    for c in cases:
        k = c["idx"]
        f = csim.Field(c["nx"], c["ny"]).upload(z[f"c{k}_in"])
        csim.apply_boundary(f, csim.bc_codes(c["bc"]), (1, 1, 1, 1), c["value"])
        assert np.array_equal(f.download(), z[f"c{k}_out"]), c


def test_reference_unit_test_assertions(csim):
    """The reference's own unit-test bodies restated against the C ABI
    (tests/simulation/unit/test_{field,diffusion,advection,boundary}.cpp)."""
    # test_diffusion.cpp:17-34
    u = np.zeros((5, 5))
    u[2, 2] = 1.0
    fu, fv = csim.Field(3, 3).upload(u), csim.Field(3, 3)
    csim.diffusion_step(fu, fv, 0.1, 0.1)
    v = fv.download()
    a = 0.01
    assert abs(v[2, 2] - (1 - 4 * a)) < 1e-12
    for (j, i) in [(2, 1), (2, 3), (1, 2), (3, 2)]:
        assert abs(v[j, i] - a) < 1e-12
    # test_advection.cpp:13-71
    h = np.zeros((10, 10))
    h[5, 5] = 1.0
    fh = csim.Field(8, 8).upload(h)
    for vx, vy, zero in [(0, 0, True), (1, 0, False), (-1, 0, False), (0, 1, False), (0, -1, False)]:
        fo = csim.Field(8, 8)
        fo.fill(0.0)
        csim.advection_step(fh, fo, float(vx), float(vy), 0.1)
        o = fo.download()
        if zero:
            assert (o[1:-1, 1:-1] == 0.0).all()
        else:
            assert o[5, 5] != 0.0
    # test_boundary.cpp:8-69
    f = np.full((5, 6), -1.0)
    f[1:-1, 1:-1] = 10.0
    ff = csim.Field(4, 3).upload(f)
    csim.apply_boundary(ff, [0, 0, 0, 0], (1, 1, 1, 1), 5.0)
    g = ff.download()
    assert (g[:, 0] == 5).all() and (g[:, -1] == 5).all() and (g[0] == 5).all() and (g[-1] == 5).all()
    f = np.full((5, 6), -1.0)
    f[1:-1, 1:-1] = np.arange(1, 4)[:, None]
    ff.upload(f)
    csim.apply_boundary(ff, [1, 1, 1, 1], (1, 1, 1, 1), 0.0)
    g = ff.download()
    assert np.array_equal(g[:, 0], g[:, 1]) and np.array_equal(g[:, -1], g[:, -2])
    assert np.array_equal(g[0], g[1]) and np.array_equal(g[-1], g[-2])
    # test_field.cpp:5-26 (layout + fill), plus Field::fill / copy / swap
    ff.fill(3.5)
    assert (ff.download() == 3.5).all()
    lay = (10.0 * np.arange(5)[:, None] + np.arange(6)[None, :])
    ff.upload(lay)
    assert np.array_equal(ff.download(), lay)
    assert np.array_equal(ff.download_interior(), lay[1:-1, 1:-1])
    other = csim.Field(4, 3)
    other.copy_from(ff)
    ff.fill(0.0)
    other.swap(ff)
    assert np.array_equal(ff.download(), lay) and (other.download() == 0).all()


def test_bad_arguments_fail_loudly(csim):
    with pytest.raises(csim.CsimError):
        csim.Field(0, 4)
    with pytest.raises(csim.CsimError):
        csim.Field(4, 4, halo=2)  # reference driver is halo==1 only
    a, b = csim.Field(4, 4), csim.Field(5, 4)
    with pytest.raises(csim.CsimError):
        csim.diffusion_step(a, b, 0.1, 0.1)
    with pytest.raises(csim.CsimError):
        csim.diffusion_step(a, a, 0.1, 0.1)
    st = csim.Stepper.single(8, 8)
    with pytest.raises(csim.CsimError):
        st.set_option("no_such_option", 1)
    st.close()


SEEDED = [
    # nx, ny, dx, dy, D, vx, vy, dt, bc, steps
    (1000, 777, 1.0, 1.0, 0.05, 0.5, 0.25, 0.1, "dnpd", 11),
    (512, 64, 1.0, 1.0, 0.2, -0.3, -0.6, 0.1, "nnnn", 9),
    (129, 130, 0.5, 0.25, 0.01, 0.3, -0.2, 0.05, "ndpn", 13),   # exact-reciprocal path
    (257, 33, 0.7, 1.3, 0.08, -0.6, 0.9, 0.1, "pdnd", 7),       # IEEE-division path
    (4096, 96, 1.0, 1.0, 1.0, 0.0, 0.0, 0.1, "pppp", 6),        # config 2 physics
    (127, 1, 1.0, 1.0, 0.1, 0.4, 0.4, 0.1, "dddd", 5),
    (1, 300, 1.0, 1.0, 0.1, 0.4, -0.4, 0.1, "nndd", 5),
    (2049, 515, 1.0, 1.0, 0.05, 0.5, 0.25, 0.1, "dddd", 8),     # config 3 physics, ragged strips
    # widths that are multiples of 128: whole strips
    (128, 5, 1.0, 1.0, 0.1, 0.3, -0.2, 0.1, "dnpd", 9),
    (256, 1, 1.0, 1.0, 0.1, -0.3, 0.2, 0.1, "nnnn", 8),
    (128, 2, 1.0, 1.0, 0.1, 0.3, 0.2, 0.1, "pppp", 7),
    (384, 130, 1.0, 1.0, 0.15, -0.4, -0.1, 0.1, "nnnn", 12),
    (1024, 200, 1.0, 1.0, 0.05, 0.5, 0.25, 0.1, "pnnp", 10),
    (640, 67, 1.0, 1.0, 0.05, 0.5, -0.25, 0.1, "ndnd", 11),
    (256, 40, 0.5, 0.25, 0.01, 0.3, -0.2, 0.05, "dnnd", 9),     # exact-reciprocal path, fused
    (128, 30, 0.7, 1.3, 0.08, -0.6, 0.9, 0.1, "npdn", 7),       # IEEE-division path, fused
    (512, 70, 1.0, 1.0, 0.1, -0.3, 0.2, 0.1, "dnpn", 10),
    (768, 9, 0.5, 0.5, 0.02, 0.2, 0.3, 0.1, "nnpp", 9),
]


@pytest.mark.parametrize("case", SEEDED, ids=[f"{c[0]}x{c[1]}_{c[8]}" for c in SEEDED])
def test_seeded_random_vs_oracle_bit_exact(csim, case):
    nx, ny, dx, dy, D, vx, vy, dt, bc, steps = case
    rng = np.random.default_rng(nx * 7919 + ny)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    # non-zero ghosts too: periodic sides must carry them unchanged through every step (Q1)
    u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = 0.25, -0.5, 0.75, -1.25
    want = u0.copy()
    ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes(bc), steps)
    for opts in [dict(variant=1), dict(variant=2), dict(variant=1, prefetch=4, rows_per_chunk=37),
                 dict(fuse=0), dict(fuse=2, rows_per_chunk=1), dict(fuse=2, rows_per_chunk=3, prefetch=1),
                 dict(fuse=2, rows_per_chunk=64, prefetch=4), dict(fuse=2, xcd_swizzle=0),
                 dict(fuse=3), dict(fuse=3, rows_per_chunk=2, prefetch=4), dict(fuse=3, rows_per_chunk=1),
                 dict(fuse=4), dict(fuse=4, rows_per_chunk=5, prefetch=4), dict(fuse=4, xcd_swizzle=0),
                 dict(fuse=4, rows_per_chunk=1),
                 dict(fuse=5), dict(fuse=6), dict(fuse=6, rows_per_chunk=3), dict(fuse=5, rows_per_chunk=1),
                 dict(fuse=6, xcd_swizzle=0, rows_per_chunk=7), dict(fuse=7), dict(fuse=7, rows_per_chunk=2),
                 dict(fuse=7, rows_per_chunk=40, xcd_swizzle=0)]:
        got = run_gpu(csim, u0, dx, dy, D, vx, vy, dt, csim.bc_codes(bc), steps, opts)
        assert np.array_equal(got, want), (opts, float(np.abs(got - want).max()))
    got = run_gpu(csim, u0, dx, dy, D, vx, vy, dt, csim.bc_codes(bc), steps, None,
                  split=[1, 2, steps - 3])
    assert np.array_equal(got, want)


def test_fuzz_random_shapes_options_and_physics_vs_oracle(csim):
    """1500 seeded random cases — any width and height from 1 cell up, any BC mix, spacing class (unit /
    power-of-two / general: the three division modes), velocity signs, step count, pass depth, chunk height,
    block map — each compared with the oracle bit for bit, ghost ring included.  The fixed lists above aim at
    known seams; this one is for the shapes nobody thought of."""
    rng = np.random.default_rng(int(os.environ.get("CSIM_FUZZ_SEED", "20260704")))   # (other seeds: extended soak runs)
    spacings = [(1.0, 1.0), (0.5, 0.25), (2.0, 0.5), (0.7, 1.3), (1.0, 0.3)]
    for case in range(1500):
        big = case % 10 == 0
        nx = int(rng.integers(1, 1500 if big else 400))
        ny = int(rng.integers(1, 900 if big else 150))
        dx, dy = spacings[int(rng.integers(0, len(spacings)))]
        D = float(rng.choice([0.0, 0.01, 0.05, 0.2]))
        vx = float(rng.choice([0.0, 0.5, -0.5, 0.25, -1.0]))
        vy = float(rng.choice([0.0, 0.25, -0.25, 0.75]))
        dt = 0.8 * min(0.1, ora.safe_dt(dx, dy, vx, vy, D)) if (D or vx or vy) else 0.1
        bc = "".join(rng.choice(list("dnp"), 4))
        steps = int(rng.integers(1, 30))
        opts = dict(fuse=int(rng.choice([-1, -1, -1, 0, 2, 3, 4, 5, 6, 7])), rows_per_chunk=int(rng.choice([0, 0, 1, 2, 5, 13, 40])),
                    xcd_swizzle=int(rng.integers(0, 2)), tail_split=int(rng.integers(0, 3)))
        u0 = np.zeros((ny + 2, nx + 2))
        u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
        u0[0, :], u0[-1, :], u0[:, 0], u0[:, -1] = rng.standard_normal(4)   # Periodic ghosts must survive
        want = u0.copy()
        ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes(bc), steps)
        split = None
        if steps >= 3 and case % 3 == 0:
            a = int(rng.integers(1, steps - 1))
            split = [a, steps - a]
        got = run_gpu(csim, u0, dx, dy, D, vx, vy, dt, csim.bc_codes(bc), steps, opts, split=split)
        assert np.array_equal(got, want), (case, nx, ny, dx, dy, D, vx, vy, bc, steps, opts, split,
                                           float(np.abs(got - want).max()))


def test_subnormal_huge_and_nonfinite_values(csim):
    """IEEE corner cases: subnormal and near-overflow magnitudes must come out bit-identical
    (fp64 denormals are not flushed on gfx950), and NaN / Inf must spread to exactly the same
    cells as on the CPU (NaN payloads are not compared)."""
    nx, ny, steps = 256, 37, 7
    rng = np.random.default_rng(77)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[1:-1, 1:65] *= 1e-310          # subnormal band
    u0[1:-1, 65:129] *= 1e-300        # products underflow into the subnormal range
    u0[1:-1, 129:193] *= 1e150        # huge but finite
    bc = "dnpd"
    for fuse in (0, 2, 4, 6):
        want = u0.copy()
        ora.run_single(want, 1.0, 1.0, 0.05, 0.5, -0.25, 0.1, ora.bc_codes(bc), steps)
        got = run_gpu(csim, u0, 1.0, 1.0, 0.05, 0.5, -0.25, 0.1, csim.bc_codes(bc), steps, dict(fuse=fuse))
        assert np.isfinite(want).all()
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), fuse
    u1 = u0.copy()
    u1[10, 40] = np.nan
    u1[20, 200] = np.inf
    u1[30, 100] = -np.inf
    want = u1.copy()
    ora.run_single(want, 1.0, 1.0, 0.05, 0.5, -0.25, 0.1, ora.bc_codes(bc), 4)
    for fuse in (0, 4):
        got = run_gpu(csim, u1, 1.0, 1.0, 0.05, 0.5, -0.25, 0.1, csim.bc_codes(bc), 4, dict(fuse=fuse))
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok], want[ok])


@pytest.mark.parametrize("dx,dy", [(1.0, 1.0), (0.5, 2.0), (0.7, 1.3)])
@pytest.mark.parametrize("fuse", [2, 4, 6, 7])
def test_fused_2c_guard_values_near_overflow(csim, dx, dy, fuse):
    """The interior body of the multi-step sweep fuses E - 2c into fma(-2, c, E), which equals the reference's
    two operations unless 2c overflows; tiles screen their inputs and fall back to the reference's own sequence.
    A field with many interior-body tiles (3+ strips, short chunks) carries patches around the critical
    magnitudes: 1e300 (below the screen: fused form, finite), 4e307 (above the screen, 2c still finite),
    1.2e308 >= 2^1023 (2c overflows in the reference: +-inf, then NaN), plus Inf and NaN cells.  The result must
    equal the oracle's: same NaN cells, every other cell bit for bit (Inf signs included) — with the fused form
    on (default) and off."""
    nx, ny = 700, 160
    steps = fuse + 3
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    dt = min(dt, csim.safe_dt(dx, dy, vx, vy, D))
    rng = np.random.default_rng(31)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    u0[30:38, 150:170] *= 1e300
    u0[60:66, 300:330] *= 4e307 / 3
    u0[90:97, 420:450] = 1.2e308 * np.sign(u0[90:97, 420:450])
    u0[120, 260] = np.inf
    u0[125, 520] = -np.inf
    u0[45, 380] = np.nan
    u0[130:134, 200:230] *= 1e-310
    want = u0.copy()
    with np.errstate(all="ignore"):
        ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes("dddd"), steps)
    assert np.isinf(want).any() and np.isnan(want).any() and np.isfinite(want).sum() > 0.5 * want.size
    for fused in (1, 0):
        st = csim.Stepper.single(nx, ny, dx, dy, csim.bc_codes("dddd"))
        for k, v in dict(fuse=fuse, rows_per_chunk=18, fused_2c=fused).items():
            st.set_option(k, v)
        st.upload(u0)
        st.run(D, dt, vx, vy, steps)
        assert st.get_option("fused_2c_active") == fused
        got = st.download()
        st.close()
        assert np.array_equal(np.isnan(got), np.isnan(want)), (fused, int((np.isnan(got) != np.isnan(want)).sum()))
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok].view(np.int64), want[ok].view(np.int64)), fused


def test_fused_2c_is_switched_off_for_wildly_unstable_parameters(csim):
    """the screen's threshold divides 2^1022 by the seventh power of the per-step growth bound; parameters whose
    bound leaves no room (dt far beyond any stability limit through the C ABI, which does not clamp) must run the
    plain form, and the result must still be the oracle's"""
    nx, ny, steps = 500, 90, 8
    rng = np.random.default_rng(5)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.standard_normal((ny, nx))
    D, vx, vy, dt = 3.0e18, 0.5, 0.25, 1.0
    want = u0.copy()
    with np.errstate(all="ignore"):
        ora.run_single(want, 1.0, 1.0, D, vx, vy, dt, ora.bc_codes("dddd"), steps)
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("dddd"))
    st.set_option("fuse", 6)
    st.set_option("rows_per_chunk", 18)
    st.upload(u0)
    st.run(D, dt, vx, vy, steps)
    assert st.get_option("fused_2c_active") == 0
    got = st.download()
    st.close()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.int64), want[ok].view(np.int64))


def test_fused_step_equals_copy_diffusion_advection(csim):
    """csim_fused_step == std::copy + diffusion_step + advection_step (main.cpp:104-107)."""
    rng = np.random.default_rng(3)
    nx, ny = 333, 211
    u = rng.standard_normal((ny + 2, nx + 2))
    fu = csim.Field(nx, ny, 1, 0.5, 2.0).upload(u)
    a, b = csim.Field(nx, ny, 1, 0.5, 2.0), csim.Field(nx, ny, 1, 0.5, 2.0)
    a.copy_from(fu)
    csim.diffusion_step(fu, a, 0.07, 0.1)
    csim.advection_step(fu, a, -0.3, 0.2, 0.1)
    b.fill(9.0)
    csim.fused_step(fu, b, 0.07, 0.1, -0.3, 0.2)
    assert np.array_equal(a.download(), b.download())
    assert a.linf_diff(b) == 0.0
    want = u.copy()
    tmp = u.copy()
    ora.diffusion_step(want, tmp, 0.5, 2.0, 0.07, 0.1)
    ora.advection_step(want, tmp, 0.5, 2.0, -0.3, 0.2, 0.1)
    assert np.array_equal(a.download(), tmp)


def test_async_snapshot_captures_the_state_at_begin(csim):
    """csim_stepper_snapshot_begin/_wait: the loop keeps stepping while the snapshot travels; the
    data handed out is the interior as of begin() (reference write_field_netcdf semantics)."""
    nx, ny = 1536, 700
    rng = np.random.default_rng(21)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.random((ny, nx))
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("dnnd"))
    st.upload(u0)
    st.run(0.05, 0.1, 0.5, 0.25, 7)
    at7 = st.download_interior()
    st.snapshot_begin()
    st.run(0.05, 0.1, 0.5, 0.25, 20)   # enqueued right behind the snapshot
    snap = st.snapshot_wait()
    assert np.array_equal(snap, at7)
    st.snapshot_begin()
    st.snapshot_begin()                # a second begin waits for the first one
    at27 = st.snapshot_wait()
    assert np.array_equal(at27, st.download_interior())
    with pytest.raises(csim.CsimError):
        st.snapshot_wait()             # nothing in flight any more
    st.close()
    want = u0.copy()
    ora.run_single(want, 1.0, 1.0, 0.05, 0.5, 0.25, 0.1, ora.bc_codes("dnnd"), 27)
    assert np.array_equal(at27, want[1:-1, 1:-1])


def test_reductions(csim):
    rng = np.random.default_rng(8)
    nx, ny = 1234, 567
    u = rng.standard_normal((ny + 2, nx + 2))
    f = csim.Field(nx, ny).upload(u)
    mn, mx = f.minmax()  # whole array, ghosts included (reference main.cpp:73-77)
    assert mn == u.min() and mx == u.max()
    s = f.sum()
    ref = float(np.sum(u[1:-1, 1:-1]))
    assert abs(s - ref) <= 1e-9 * np.abs(u[1:-1, 1:-1]).sum()
    g = csim.Field(nx, ny).upload(u + 0.0)
    assert f.linf_diff(g) == 0.0
    u2 = u.copy()
    u2[100, 200] += 0.125
    g.upload(u2)
    assert f.linf_diff(g) == abs(u[100, 200] - u2[100, 200])


def test_device_gaussian_matches_host_formula(csim):
    """NEXT-1 row: IC on device (reference src/init.cpp:12-33); exp() may differ by an ulp."""
    nx, ny = 640, 384
    st = csim.Stepper.single(nx, ny, 1.0, 1.0)
    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
    got = st.download()
    st.close()
    want = ora.gaussian_global(nx, ny)
    assert (got[0] == 0).all() and (got[:, 0] == 0).all()
    assert np.abs(got - want).max() <= 4 * np.finfo(float).eps


# ---- BASELINE.json full sizes: domain-of-dependence windows -----------------------------------
def _window_check(csim, nx, ny, D, vx, vy, dt, bc, steps, opts, nwin, seed):
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes(bc))
    for k, v in (opts or {}).items():
        st.set_option(k, v)
    rng = np.random.default_rng(seed)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.random((ny, nx))
    st.upload(u0)
    st.set_option("profile", 1)
    st.run(D, dt, vx, vy, steps)
    got = st.download()
    # which kernel advanced the field, and with which chunk height
    depth_used = max((1, 2, 3, 4, 5, 6, 7), key=lambda t: t * st.kernel_time(t)[1])
    rows_used = st.get_option("last_rows") if depth_used >= 2 else (opts or {}).get("rows_per_chunk", 64)
    st.close()
    W = 96
    anchors = [(1, 1), (nx - W + 1, 1), (1, ny - W + 1), (nx - W + 1, ny - W + 1)]
    # strip seams of the overlapped-strip kernel lie at multiples of OverlapGeom<T>::STRIDE =
    # 128 - 4 * ceil(T / 2) output columns (112 at T = 7; 116 at T = 5, 6; 120 at T = 3, 4; 124 at T = 2), chunk
    # seams at 1 + k * (rows the launch used): centre windows on both
    stride = {2: 124, 3: 120, 4: 120, 5: 116, 6: 116, 7: 112}.get(depth_used, 128)
    nstrips = (nx + stride - 1) // stride
    nchunks = max(1, (ny + rows_used - 1) // rows_used) if rows_used > 0 else 1
    for k in range(nwin):
        if k < len(anchors):
            i0, j0 = anchors[k]
        else:
            i0 = (1 + stride * int(rng.integers(1, max(2, nstrips))) - W // 2 + int(rng.integers(-3, 4))) if k % 2 \
                else int(rng.integers(1, nx - W))
            j0 = (1 + rows_used * int(rng.integers(1, max(2, nchunks))) - W // 2 + int(rng.integers(-2, 3))) \
                if (k % 3 == 0 and rows_used > 0) else int(rng.integers(1, ny - W))
            i0, j0 = max(1, min(i0, nx - W + 1)), max(1, min(j0, ny - W + 1))
        # expanded window, clipped at the physical edges (where the real BC applies)
        a0, a1 = max(1, i0 - steps), min(nx, i0 + W - 1 + steps)
        b0, b1 = max(1, j0 - steps), min(ny, j0 + W - 1 + steps)
        sub = u0[b0 - 1:b1 + 2, a0 - 1:a1 + 2].copy()
        # sides cut inside the domain get a frozen (Periodic = no-op) ring: wrong values there
        # contaminate at most `steps` cells, which lie outside the compared window
        codes = ora.bc_codes(bc)
        sub_bc = [codes[0] if a0 == 1 else ora.PERIODIC, codes[1] if a1 == nx else ora.PERIODIC,
                  codes[2] if b0 == 1 else ora.PERIODIC, codes[3] if b1 == ny else ora.PERIODIC]
        ora.run_single(sub, 1.0, 1.0, D, vx, vy, dt, sub_bc, steps)
        wi, wj = i0 - (a0 - 1), j0 - (b0 - 1)
        want = sub[wj:wj + W, wi:wi + W]
        have = got[j0:j0 + W, i0:i0 + W]
        assert np.array_equal(have, want), (k, i0, j0, float(np.abs(have - want).max()))
    return got


@pytest.mark.parametrize("fuse", [-1, 2, 3, 4, 5])
def test_full_size_config2_4096_diffusion_periodic(csim, fuse):
    got = _window_check(csim, 4096, 4096, 1.0, 0.0, 0.0, 0.1, "pppp", 14, dict(fuse=fuse), 12, 42)
    assert np.isfinite(got).all()


def test_full_size_config3_8192_dirichlet(csim):
    _window_check(csim, 8192, 8192, 0.05, 0.5, 0.25, 0.1, "dddd", 8, None, 12, 43)


def test_full_size_16384_windows(csim):
    _window_check(csim, 16384, 16384, 0.05, 0.5, 0.25, 0.1, "dnnd", 8, None, 10, 44)


@pytest.mark.parametrize("bc", ["dddd", "dnnd"])
@pytest.mark.parametrize("ic", ["gaussian", "random"])
def test_full_field_16384_exactly_what_bench_times(csim, bc, ic):
    """The WHOLE 16384 x 16384 field (ghost ring included) against the oracle, bit for bit, through the
    very launches bench.py times: a 36-step run (>= 4 x depth, so the on-device chunk-height trial fires
    and the six passes are k_sweepO_dpp<T=6> with the tuned rows), then 20 more steps on the same
    stepper (7 + 7 + 6: two passes of k_sweepO_dpp<T=7> and one of <T=6> with the tuned rows re-snapped —
    the schedule of the driver's `bench.py --steps 20`), then 10 more (5 + 5).  Oracle: 16 tiles / 16 threads of oracle/cpu_stepper.c
    (= the reference under mpirun -np 16), reassembled with its physical ghost lines."""
    n = 16384
    D, vx, vy, dt = 0.05, 0.5, 0.25, 0.1      # bench.py PHYS
    if ic == "gaussian":
        u0 = ora.gaussian_global(n, n)
    else:
        u0 = np.zeros((n + 2, n + 2))
        u0[1:-1, 1:-1] = np.random.default_rng(160).random((n, n))
    w = ora.World(16, n, n)
    w.scatter(np.ascontiguousarray(u0[1:-1, 1:-1]))
    st = csim.Stepper.single(n, n, 1.0, 1.0, csim.bc_codes(bc))
    st.upload(u0)
    del u0
    st.set_option("profile", 1)
    done = 0
    for steps, passes in [(36, {6: 6}), (20, {7: 2, 6: 1}), (10, {5: 2})]:
        st.reset_timers()
        st.run(D, dt, vx, vy, steps)
        got = st.download()
        ran = {t: st.kernel_time(t)[1] for t in range(1, 8) if st.kernel_time(t)[1]}
        assert ran == passes, ran
        tuned = st.get_option("tuned_rows")
        assert tuned > 0, "the chunk-height trial did not run"
        last, depth = st.get_option("last_rows"), min(passes)   # deep passes first: the last launch is the shallowest
        assert tuned <= last < tuned + 6 and (last + 2 * (depth - 1)) % 6 == 0, (tuned, last)
        w.run(D, vx, vy, dt, ora.bc_codes(bc), steps, threads=16)
        want = w.gather_full()
        done += steps
        same = np.array_equal(got, want)
        assert same, (bc, ic, done, float(np.abs(got - want).max()), int((got != want).sum()))
        del got, want
    st.close()


def _full_field(csim, nx, ny, D, vx, vy, bc, runs, seed, tiles=16):
    """whole field incl. the ghost ring vs `tiles` oracle tiles on as many threads, after each run of `runs`"""
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = np.random.default_rng(seed).random((ny, nx))
    w = ora.World(tiles, nx, ny)
    w.scatter(np.ascontiguousarray(u0[1:-1, 1:-1]))
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes(bc))
    st.upload(u0)
    del u0
    for steps in runs:
        st.run(D, 0.1, vx, vy, steps)
        got = st.download()
        w.run(D, vx, vy, 0.1, ora.bc_codes(bc), steps, threads=tiles)
        want = w.gather_full()
        assert np.array_equal(got, want), (nx, ny, bc, steps, float(np.abs(got - want).max()))
        del got, want
    st.close()


def test_full_field_config2_4096_and_config3_8192(csim):
    """BASELINE configs[1] (4096^2 diffusion-only, periodic) and configs[2] (8192^2, Dirichlet, both upwind
    branches): every cell and ghost after a tuned-chunk run (30 steps = 5 x T=6) plus a 7-step tail (4 + 3)"""
    _full_field(csim, 4096, 4096, 1.0, 0.0, 0.0, "pppp", [30, 7], 201)
    _full_field(csim, 8192, 8192, 0.05, 0.5, 0.25, "dddd", [30, 7], 202)
    _full_field(csim, 8192, 8192, 0.05, -0.5, -0.25, "dddd", [12], 203)


def test_full_field_config5_32768_neumann(csim):
    """BASELINE configs[4]'s grid, 32768 x 32768 all-Neumann (2 x 8.6 GB on the device), as ONE field: the
    reference is decomposition-invariant, so this is also what the 4 x 2 run must assemble to.  Every cell
    and ghost after 30 steps (chunk-height trial + five T=6 passes) vs 16 oracle tiles."""
    _full_field(csim, 32768, 32768, 0.05, 0.5, 0.25, "nnnn", [30], 204)


def test_full_size_config5_32768_neumann(csim):
    """BASELINE configs[4]: 32768 x 32768, all-Neumann (2 x 8.6 GB on the device).  The reference is
    decomposition-invariant, so the single-GPU field is what the 4 x 2 run must give as well."""
    _window_check(csim, 32768, 32768, 0.05, 0.5, 0.25, 0.1, "nnnn", 8, None, 8, 45)


def test_long_run_all_schedules_agree_bitwise(csim):
    """1500 steps at 2048^2: six-, four-, two-step passes, single steps and the LDS kernel must
    leave bit-identical fields (same arithmetic per cell, whatever the schedule), mass conserved
    (Neumann walls), maximum principle respected."""
    nx = ny = 2048
    steps = 1500
    outs = {}
    sums = {}
    for name, opts in [("fuse6", dict(fuse=6)), ("fuse4", dict(fuse=4)), ("fuse2", dict(fuse=2)),
                       ("fuse0", dict(fuse=0)), ("lds", dict(fuse=0, variant=2))]:
        st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes("nnnn"))
        for k, v in opts.items():
            st.set_option(k, v)
        st.init_gaussian(1.0, 0.05, 0.5, 0.5)
        if name == "fuse6":
            mass0, (mn0, mx0) = st.sum(), st.minmax()
        # uneven call pattern on purpose
        st.run(0.2, 0.1, 0.5, -0.25, 1)
        st.run(0.2, 0.1, 0.5, -0.25, 998)
        st.run(0.2, 0.1, 0.5, -0.25, steps - 999)
        outs[name] = st.download()
        sums[name] = (st.sum(), st.minmax())
        st.close()
    for name in outs:
        assert np.array_equal(outs[name], outs["fuse0"]), name
    mass1, (mn1, mx1) = sums["fuse6"]
    assert abs(mass1 - mass0) <= 1e-12 * abs(mass0)
    assert mn1 >= mn0 - 1e-15 and mx1 < mx0


def test_physics_sanity_like_reference_integration_tests(csim):
    """reference tests/simulation/integration/integration_{diffusion,advection}.cpp: the peak of
    a diffusing hotspot decreases and stays >= 0; an advected hotspot's centre of mass moves by
    ~v*t with mass kept within 5 %."""
    nx = ny = 64
    u0 = ora.gaussian_global(nx, ny)
    st = csim.Stepper.single(nx, ny)
    st.upload(u0)
    st.run(1.0, 0.1, 0.0, 0.0, 9)
    d = st.download_interior()
    assert d.max() < u0.max() and (d >= 0).all()
    st.upload(u0)
    st.run(0.0, 1.0, 1.0, 0.0, 5)  # vx=1, dt=1 (CFL limit), 5 steps
    a = st.download_interior()
    st.close()
    x = np.arange(nx)[None, :]
    com0 = (u0[1:-1, 1:-1] * x).sum() / u0[1:-1, 1:-1].sum()
    com1 = (a * x).sum() / a.sum()
    assert abs((com1 - com0) - 5.0) <= 1.0
    assert abs(a.sum() - u0.sum()) <= 0.05 * u0.sum()
