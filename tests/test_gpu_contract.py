"""Option "contract" = 1 (opt-in, off by default): the cell update evaluated as the 5-point FMA stencil
a0 c + aW W + aE E + aS S + aN N instead of the reference's 15 non-FMA operations.  NOT bit-identical by
design; the bar here is the north-star tolerance, L_inf vs the CPU reference < 1e-10, written below.  The
default (contract = 0) must stay bit-identical — that is what every other GPU test checks."""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import cpu_oracle as ora

pytestmark = pytest.mark.gpu

TOL = 1e-10   # BASELINE.json north_star: "L-infinity error vs CPU reference < 1e-10"


@pytest.fixture(scope="module")
def csim():
    pkg = load_package()
    pkg.lib()
    pkg.set_device(0)
    return pkg


@pytest.mark.parametrize("bc", ["dddd", "nnnn", "pppp", "dnpd", "ndnp", "pnnd"])
def test_contract_within_tolerance_after_1000_steps_2048(csim, bc):
    n, steps = 2048, 1000
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    u0 = ora.gaussian_global(n, n, sigma_frac=0.08, xc_frac=0.45, yc_frac=0.6)
    rng = np.random.default_rng(12)
    u0[1:-1, 1:-1] += 0.05 * rng.standard_normal((n, n))     # every cell carries rounding-sensitive data
    w = ora.World(16, n, n)
    w.scatter(np.ascontiguousarray(u0[1:-1, 1:-1]))
    w.run(D, vx, vy, dt, ora.bc_codes(bc), steps, threads=16)
    want = w.gather_full()
    errs = {}
    for fuse in (-1, 0, 4):
        st = csim.Stepper.single(n, n, 1.0, 1.0, csim.bc_codes(bc))
        st.set_option("contract", 1)
        st.set_option("fuse", fuse)
        st.upload(u0)
        st.run(D, dt, vx, vy, 1)
        st.run(D, dt, vx, vy, steps - 1)
        got = st.download()
        st.close()
        errs[fuse] = float(np.abs(got - want).max())
        assert errs[fuse] < TOL, (bc, fuse, errs)
    # contraction changes bits (otherwise the option would be pointless) but only in the last places
    assert 0.0 < max(errs.values()) < 1e-12, errs
    print(f"contract=1 L_inf vs oracle after {steps} steps at {n}^2, bc={bc}: {errs}")


def test_contract_other_spacings_and_signs(csim):
    """non-unit dx, dy (coefficients absorb the divisions) and all four upwind direction pairs"""
    nx, ny, steps = 777, 515, 200
    rng = np.random.default_rng(3)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.random((ny, nx))
    for (dx, dy, vx, vy) in [(0.5, 0.25, 0.3, -0.2), (0.7, 1.3, -0.6, 0.9), (1.0, 1.0, -0.5, -0.25), (2.0, 1.0, 0.4, 0.1)]:
        D = 0.02
        dt = 0.5 * ora.safe_dt(dx, dy, vx, vy, D)
        want = u0.copy()
        ora.run_single(want, dx, dy, D, vx, vy, dt, ora.bc_codes("dnnd"), steps)
        st = csim.Stepper.single(nx, ny, dx, dy, csim.bc_codes("dnnd"))
        st.set_option("contract", 1)
        st.upload(u0)
        st.run(D, dt, vx, vy, steps)
        got = st.download()
        # the same stepper switched back must be bit-identical again
        st.set_option("contract", 0)
        st.upload(u0)
        st.run(D, dt, vx, vy, steps)
        exact = st.download()
        st.close()
        assert np.abs(got - want).max() < TOL, (dx, dy, vx, vy)
        assert np.array_equal(exact, want)


def test_contract_on_the_self_linked_torus(csim):
    """multi-rank path (deep faces over RCCL, frame / bulk split) with contracted arithmetic: the values a
    rank sends are the values its neighbour would compute, so the torus still matches a serial run of the
    same arithmetic bit for bit, and the oracle within the tolerance"""
    from test_gpu_comm import self_neighbor_decomp, torus_oracle
    nx, ny, steps = 1160, 300, 30
    D, vx, vy, dt = 0.05, 0.5, -0.25, 0.1
    rng = np.random.default_rng(9)
    u0 = np.zeros((ny + 2, nx + 2))
    u0[1:-1, 1:-1] = rng.random((ny, nx))
    want = torus_oracle(u0, 1.0, 1.0, D, vx, vy, dt, steps)
    outs = []
    for opts in (dict(fuse=-1, overlap=1), dict(fuse=0, overlap=0)):
        st = csim.Stepper(self_neighbor_decomp(csim, nx, ny, (1, 1, 1, 1)), 1.0, 1.0, csim.bc_codes("dddd"))
        st.comm_init(csim.comm_unique_id())
        st.set_option("contract", 1)
        for k, v in opts.items():
            st.set_option(k, v)
        st.upload(u0)
        st.run(D, dt, vx, vy, steps)
        outs.append(st.download_interior())
        st.close()
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0] - want[1:-1, 1:-1]).max() < TOL
