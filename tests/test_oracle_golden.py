"""The CPU restatement (oracle/cpu_stepper.c) against the golden vectors generated from the
compiled reference objects (oracle/make_golden.py).  Bit-exact: np.array_equal, no tolerance.
This is what pins the oracle; the GPU parity tests then compare the HIP path to the oracle."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import cpu_oracle as ora

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RUN_FILES = sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def with_ghosts(interior):
    ny, nx = interior.shape
    f = np.zeros((ny + 2, nx + 2))
    f[1:-1, 1:-1] = interior
    return f


@pytest.mark.parametrize("path", RUN_FILES, ids=[os.path.basename(p)[:-4] for p in RUN_FILES])
def test_run_single_tile_bit_exact(path):
    z, m = _load(path)
    dt = float(z["dt_effective"])
    assert dt == min(m["dt"], ora.safe_dt(m["dx"], m["dy"], m["vx"], m["vy"], m["D"]))
    u = with_ghosts(z["u0"])
    ora.run_single(u, m["dx"], m["dy"], m["D"], m["vx"], m["vy"], dt, ora.bc_codes(m["bc"]),
                   m["steps"])
    assert np.array_equal(u[1:-1, 1:-1], z["u_final"])
    # full local array of the 1-rank reference run, ghosts and corners included
    assert np.array_equal(u, z["local_np1_rank0"])


@pytest.mark.parametrize("path", RUN_FILES, ids=[os.path.basename(p)[:-4] for p in RUN_FILES])
def test_run_multi_tile_bit_exact(path):
    z, m = _load(path)
    dt = float(z["dt_effective"])
    for p in m["ranks"]:
        w = ora.World(p, m["nx"], m["ny"], m["dx"], m["dy"])
        w.scatter(z["u0"])
        w.run(m["D"], m["vx"], m["vy"], dt, ora.bc_codes(m["bc"]), m["steps"],
              threads=min(p, 4))
        assert np.array_equal(w.gather(), z["u_final"]), p
        # the tiles reassembled with the ghost lines of their physical sides == the full local array
        # of the reference's own 1-rank run (what the full-size GPU parity tests compare against)
        assert np.array_equal(w.gather_full(), z["local_np1_rank0"]), p
        if p == m["ranks"][-1]:  # the bounds-checked accessor flavour (CPU-baseline variant): same bits
            wc = ora.World(p, m["nx"], m["ny"], m["dx"], m["dy"])
            wc.scatter(z["u0"])
            wc.run(m["D"], m["vx"], m["vy"], dt, ora.bc_codes(m["bc"]), m["steps"], threads=2, checked=True)
            assert np.array_equal(wc.gather_full(), z["local_np1_rank0"]), p
        for r in range(p):
            got, want = w.tile(r), z[f"local_np{p}_rank{r}"]
            assert got.shape == want.shape
            # everything but the four corner ghosts (undefined in the reference)
            mask = np.ones(got.shape, bool)
            mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False
            if p == 1:
                mask[:] = True
            assert np.array_equal(got[mask], want[mask]), (p, r)


def test_unit_steps_bit_exact():
    z, cases = _load(os.path.join(GOLDEN, "unit_steps.npz"))
    for c in cases:
        k = c["idx"]
        u = z[f"c{k}_u"].copy()
        o = z[f"c{k}_o"].copy()
        if c["op"] == "diffusion":
            ora.diffusion_step(u, o, c["dx"], c["dy"], c["D"], c["dt"])
        else:
            ora.advection_step(u, o, c["dx"], c["dy"], c["vx"], c["vy"], c["dt"])
        assert np.array_equal(o, z[f"c{k}_out"]), c
        assert np.array_equal(u, z[f"c{k}_u"])


def test_reference_known_answer_impulse():
    """reference tests/simulation/unit/test_diffusion.cpp:17-34 (tolerance 1e-12 there)."""
    u = np.zeros((5, 5))
    u[2, 2] = 1.0
    v = np.zeros((5, 5))
    ora.diffusion_step(u, v, 1.0, 1.0, 0.1, 0.1)
    a = 0.1 * 0.1
    assert abs(v[2, 2] - (1 - 4 * a)) < 1e-12
    for (j, i) in [(2, 1), (2, 3), (1, 2), (3, 2)]:
        assert abs(v[j, i] - a) < 1e-12


def test_reference_advection_zero_velocity():
    """reference tests/simulation/unit/test_advection.cpp:13-23."""
    u = np.zeros((10, 10))
    u[5, 5] = 1.0
    out = np.zeros((10, 10))
    ora.advection_step(u, out, 1.0, 1.0, 0.0, 0.0, 0.1)
    assert (out[1:-1, 1:-1] == 0.0).all()
    for vx, vy in [(1, 0), (-1, 0), (0, 1), (0, -1)]:
        out[:] = 0
        ora.advection_step(u, out, 1.0, 1.0, float(vx), float(vy), 0.1)
        assert out[5, 5] != 0.0


def test_boundary_bit_exact():
    z, cases = _load(os.path.join(GOLDEN, "boundary.npz"))
    for c in cases:
        k = c["idx"]
        f = z[f"c{k}_in"].copy()
        ora.apply_boundary(f, ora.bc_codes(c["bc"]), (1, 1, 1, 1), c["value"])
        assert np.array_equal(f, z[f"c{k}_out"]), c


def test_decomp_table_matches_mpi():
    z, m = _load(os.path.join(GOLDEN, "decomp_table.npz"))
    for (nx, ny) in m["grids"]:
        for p in m["sizes"]:
            want = z[f"g{nx}x{ny}_np{p}"]
            for r in range(p):
                d = ora.decomp(p, r, nx, ny)
                assert list(d.values()) == list(want[r]), (nx, ny, p, r)


def test_safe_dt_table():
    t = np.load(os.path.join(GOLDEN, "safe_dt.npz"))["table"]
    for dx, dy, vx, vy, D, want in t:
        assert ora.safe_dt(dx, dy, vx, vy, D) == want


def test_periodic_is_noop_equals_dirichlet_zero():
    """Periodic leaves the (zero) ghosts alone, i.e. behaves as Dirichlet(0)."""
    rng = np.random.default_rng(5)
    a = with_ghosts(rng.random((12, 10)))
    b = a.copy()
    ora.run_single(a, 1.0, 1.0, 0.1, 0.3, -0.2, 0.1, ora.bc_codes("pppp"), 6)
    ora.run_single(b, 1.0, 1.0, 0.1, 0.3, -0.2, 0.1, ora.bc_codes("dddd"), 6)
    assert np.array_equal(a, b)
