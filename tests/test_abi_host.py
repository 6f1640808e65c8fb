"""CPU-side checks of the product's host logic and of the C-ABI library (no GPU needed):
the library loads, exports every symbol include/csim.h declares, and its MPI-free
decomposition / safe_dt agree with the real MPI library / the reference header (goldens)."""
import json
import os

import numpy as np
import pytest

from __graft_entry__ import load_package

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def csim():
    pkg = load_package()
    pkg.build()
    return pkg


def test_library_exports_every_declared_symbol(csim):
    lib = csim.lib()
    names = csim.declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    assert lib.csim_abi_version() == 1


def test_decomp_matches_mpi_tables(csim):
    z = np.load(os.path.join(GOLDEN, "decomp_table.npz"), allow_pickle=False)
    m = json.loads(str(z["meta"]))
    for (nx, ny) in m["grids"]:
        for p in m["sizes"]:
            want = z[f"g{nx}x{ny}_np{p}"]
            for r in range(p):
                d = csim.decomp_init(p, r, nx, ny).as_dict()
                assert list(d.values()) == list(want[r]), (nx, ny, p, r)


def test_decomp_rejects_bad_arguments(csim):
    with pytest.raises(csim.CsimError):
        csim.decomp_init(4, 4, 16, 16)
    with pytest.raises(csim.CsimError):
        csim.decomp_init(0, 0, 16, 16)
    with pytest.raises(csim.CsimError):
        csim.decomp_init(1, 0, 0, 16)
    with pytest.raises(csim.CsimError):
        csim.decomp_init(8, 0, 2, 2)  # more ranks than cells along x


def test_safe_dt_matches_reference_header(csim):
    t = np.load(os.path.join(GOLDEN, "safe_dt.npz"))["table"]
    for dx, dy, vx, vy, D, want in t:
        assert csim.safe_dt(dx, dy, vx, vy, D) == want
    assert csim.safe_dt(1, 1, 0, 0, 0) == float("inf")


def test_bc_aliases(csim):
    # reference src/io.cpp:35-44
    assert csim.bc_from_string("Fixed") == csim.DIRICHLET
    assert csim.bc_from_string("noflux") == csim.NEUMANN
    assert csim.bc_from_string("zero-flux") == csim.NEUMANN
    assert csim.bc_from_string("period") == csim.PERIODIC
    with pytest.raises(RuntimeError):
        csim.bc_from_string("robin")


def test_no_cpu_fallback_without_device(csim):
    """On a box without a GPU the compute entry points must fail loudly, never fall back."""
    try:
        n = csim.device_count()
    except csim.CsimError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(csim.CsimError):
        csim.Field(8, 8)
    with pytest.raises(csim.CsimError):
        csim.Stepper.single(8, 8)


def test_pass_schedule_is_a_pure_function_with_the_documented_properties():
    """csim_pass_schedule: the split of a run into HBM passes of 2..7 time steps (host arithmetic, no GPU).
    Every rank derives it from (nsteps, smallest tile, fuse) alone, so the checks here hold for all ranks."""
    pkg = load_package()
    ps = pkg.pass_schedule
    assert ps(0) == [] and ps(1) == [1]
    assert ps(20) == [7, 7, 6]                    # the driver's `bench.py --steps 20`: three launches, not 4 x 5
    assert ps(36) == [6] * 6 and ps(12) == [6, 6] and ps(13) == [7, 6] and ps(10) == [5, 5]
    long = ps(1000)
    assert sum(long) == 1000 and len(long) in (166, 167) and set(long) <= {4, 5, 6, 7} and long[:150] == [6] * 150
    for cap in (1, 2, 3, 4, 5, 6, 7, 100):
        for k in list(range(0, 130)) + [997, 1000, 1001, 4099]:
            for fuse in (-1, 0, 2, 3, 6, 7):
                plan = ps(k, cap, fuse)
                assert sum(plan) == k, (cap, k, fuse)
                limit = 1 if fuse in (0, 1) else min(cap, 7 if fuse < 0 else fuse)
                assert all(1 <= t <= max(1, limit) for t in plan), (cap, k, fuse, plan)
                if limit >= 3 and k >= 2:
                    assert 1 not in plan, (cap, k, fuse, plan)      # never a single-step pass when avoidable
                if limit == 2:
                    assert plan.count(1) <= 1
                assert plan == sorted(plan[:len(plan)], reverse=True) or fuse >= 0 or k > 48, (cap, k, plan)
    import ctypes
    n = ctypes.c_long(0)                            # a billion steps: planned without materialising anything of that size
    assert pkg.lib().csim_pass_schedule(10 ** 9, 1 << 30, 0, -1, None, 0, ctypes.byref(n)) == 0 and 166666660 <= n.value <= 166666670
    big = 16384 * 16384                             # tiles of >= 2e8 cells prefer depth 7 (0.9 % cheaper per step there)
    assert ps(1000, tile_cells=big) == [7] * 142 + [6] and ps(20, tile_cells=big) == [7, 7, 6]
    assert ps(36, tile_cells=big) == [6] * 6 and ps(35, tile_cells=big) == [7] * 5 and ps(12, tile_cells=big) == [6, 6]
    assert ps(20, tile_cells=4096 * 8192) == [5, 5, 5, 5]   # depth 7 costs 8-10 % more per step on the 8-GPU tile
    # the diffusion-only flavour (v == 0, csim_pass_schedule_for): HBM-bound, 7 steps per pass at every tile size
    for cells in (0, 10 ** 6, 4096 * 4096, 4096 * 8192, big):
        assert ps(20, tile_cells=cells, diffusion_only=True) == [7, 7, 6] and ps(700, tile_cells=cells, diffusion_only=True) == [7] * 100
        for k in list(range(0, 60)) + [997, 1001]:
            for cap in (1, 2, 3, 5, 7, 100):
                plan = ps(k, cap, -1, cells, True)
                assert sum(plan) == k and all(1 <= t <= max(1, min(cap, 7)) for t in plan), (cells, k, cap, plan)
                assert cap < 3 or k < 2 or 1 not in plan
    assert pkg.lib().csim_pass_schedule_for(10, 1 << 30, 0, 9, 1, None, 0, ctypes.byref(n)) != 0   # fuse out of range
    assert ps(100, tile_cells=512 * 512)[:12] == [4] * 12 and ps(100, tile_cells=8192 * 8192)[:8] == [6] * 8
    with pytest.raises(pkg.CsimError):
        ps(5, 8, 9)
