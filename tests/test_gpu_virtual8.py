"""The 4 x 2 (and 3 x 2, 2 x 2, 2 x 1) process grids through the HIP path on ONE GPU: every rank is a real
csim_stepper handle on its own tile, all in this process, faces routed host-side (tests/virtual_ranks.py).

* the reference's own `mpirun -np 8` / `-np 6` goldens (tests/golden/run_*.npz with 8- or 6-rank data: per-rank
  local arrays incl. ghost lines, global result) at every pass depth 1..7;
* BASELINE configs[3]'s grid (16384^2) on 2 x 2 and on the 8-GPU topology 4 x 2, and configs[4] as specified —
  32768^2, 4 x 2, all-Neumann, every tile loaded from a NetCDF (CDF-5) initial-condition file through
  read_netcdf_window — tile by tile, physical ghost lines included, against the ORACLE (16 tiles / threads of
  oracle/cpu_stepper.c, pinned to the compiled reference), not against another HIP run.

Not covered here: the transport (RCCL between distinct GPUs) — bench.py's parity preflight does that on the first
multi-GPU node it meets.  Reference: src/decomp.cpp:13-33 (process grid), src/halo.cpp:6-50 (what travels)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from test_multirank_gloo import cases_with
from virtual_ranks import VirtualRanks, physical_mask, tile_mask

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "driver")
TOOL = os.path.join(DRV, "csim_hosttool")


@pytest.fixture(scope="module")
def csim():
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.lib()
    pkg.set_device(0)
    return pkg


@pytest.fixture(scope="module")
def ora():
    from oracle import cpu_oracle
    return cpu_oracle


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("world", [6, 8])
def test_virtual_ranks_match_the_reference_goldens(csim, world, depth):
    cases = cases_with(world)
    assert cases and (world != 8 or len(cases) >= 5)
    for path in cases:
        z = np.load(path, allow_pickle=False)
        m = json.loads(str(z["meta"]))
        dt = float(z["dt_effective"])
        vr = VirtualRanks(csim, world, m["nx"], m["ny"], m["dx"], m["dy"], csim.bc_codes(m["bc"]), fuse=max(depth, 0))
        try:
            for r, dec in enumerate(vr.decs):
                assert list(dec.as_dict().values()) == list(z[f"decomp_np{world}"][r]), "decomposition differs from MPI's"
            vr.upload_global(z["u0"])
            vr.advance(m["D"], dt, m["vx"], m["vy"], m["steps"], depth=depth)
            glob = np.zeros((m["ny"], m["nx"]))
            for r, dec in enumerate(vr.decs):
                got, want = vr.download(r), z[f"local_np{world}_rank{r}"]
                mask = tile_mask(dec)
                assert np.array_equal(got[mask], want[mask]), (os.path.basename(path), world, depth, r,
                                                               float(np.abs(got - want)[mask].max()))
                glob[dec.y_offset:dec.y_offset + dec.ny_local, dec.x_offset:dec.x_offset + dec.nx_local] = got[1:-1, 1:-1]
            assert np.array_equal(glob, z["u_final"]), (os.path.basename(path), world, depth)
            # the per-rank checksums add up to the checksum of the global field (csim_stepper_checksum)
            assert vr.checksum() == csim.checksum_host(z["u_final"]), (os.path.basename(path), world, depth)
        finally:
            vr.close()


def _oracle_full(ora, u0_interior, bc, D, vx, vy, dt, steps):
    ny, nx = u0_interior.shape
    w = ora.World(16, nx, ny)
    w.scatter(u0_interior)
    w.run(D, vx, vy, dt, ora.bc_codes(bc), steps, threads=16)
    return w.gather_full()


def _compare_tiles(vr, want):
    for r, dec in enumerate(vr.decs):
        got = vr.download(r)
        ref = want[dec.y_offset:dec.y_offset + dec.ny_local + 2, dec.x_offset:dec.x_offset + dec.nx_local + 2]
        mask = physical_mask(dec)
        same = np.array_equal(got[mask], ref[mask])
        assert same, (r, list(dec.coords), float(np.abs(got - ref)[mask].max()), int(((got != ref) & mask).sum()))
        del got


@pytest.mark.parametrize("nx,ny,bc,steps,world", [(16384, 16384, "dddd", 14, 8), (16384, 16384, "dnnd", 9, 4),
                                                  (16384, 16384, "dddd", 8, 2)])
def test_full_size_decomposed_runs_vs_the_oracle(csim, ora, nx, ny, bc, steps, world):
    """BASELINE configs[3]'s grid on the 8-GPU process grid (4 x 2: tiles 4096 x 8192), on 2 x 2 (8192^2) and on
    2 x 1: every tile of the decomposed HIP run — interior and the ghost lines of its physical sides — equals the
    same region of the oracle's field after `steps` steps (fused passes of depth 7 + single steps)."""
    D, vx, vy, dt = 0.05, 0.5, 0.25, 0.1
    u0 = np.random.default_rng(1000 + world).random((ny, nx))
    vr = VirtualRanks(csim, world, nx, ny, 1.0, 1.0, csim.bc_codes(bc), fuse=7)
    try:
        assert [vr.decs[0].dims[0], vr.decs[0].dims[1]] == {8: [4, 2], 4: [2, 2], 2: [2, 1]}[world]
        vr.upload_global(u0)
        want = _oracle_full(ora, u0, bc, D, vx, vy, dt, steps)
        del u0
        vr.advance(D, dt, vx, vy, steps)
        _compare_tiles(vr, want)
    finally:
        vr.close()


def test_config5_32768_on_4x2_all_neumann_tiles_from_a_netcdf_ic_file(csim, ora, tmp_path):
    """BASELINE configs[4] as specified: 32768 x 32768, 2 x 4 GPUs' worth of tiles (MPI_Dims_create gives 4 x 2:
    8192 x 16384 each), all-Neumann, initial condition from a NetCDF file — every virtual rank loads ITS block with
    read_netcdf_window (the per-rank start/count of reference src/io.cpp:402-418), nine steps (one pass of seven +
    two single steps), every tile incl. its physical ghost lines vs the oracle's single field."""
    n, steps, bc = 32768, 9, "nnnn"
    D, vx, vy, dt = 0.05, 0.5, 0.25, 0.1
    free = os.statvfs(tmp_path).f_bavail * os.statvfs(tmp_path).f_frsize
    if free < 10 * 2**30:
        pytest.skip(f"needs 8.6 GB of scratch disk for the IC file, {free / 2**30:.1f} GiB free in {tmp_path}")
    subprocess.run(["make", "-s", "-C", DRV, "csim_hosttool"], check=True)
    u0 = np.random.default_rng(325).random((n, n))
    # the IC file: header by the product's writer (open_netcdf_parallel / close, zero records), the one record
    # appended here in the container's big-endian layout, record count patched (CDF-5: 64-bit numrecs at byte 4)
    nc = tmp_path / "ic.nc"
    subprocess.run([TOOL, "nc-write", str(nc), os.devnull, "0", f"--nx={n}", f"--ny={n}"], check=True)
    with open(nc, "r+b") as f:
        assert f.read(4) == b"CDF\x05"
        f.seek(0, os.SEEK_END)
        for j0 in range(0, n, 1024):
            u0[j0:j0 + 1024].astype(">f8").tofile(f)
        f.seek(4)
        f.write(struct.pack(">q", 1))
    vr = VirtualRanks(csim, 8, n, n, 1.0, 1.0, csim.bc_codes(bc), fuse=7)
    try:
        assert [vr.decs[0].dims[0], vr.decs[0].dims[1]] == [4, 2]

        def tile(r, dec):
            out = tmp_path / "tile.bin"
            res = subprocess.run([TOOL, "nc-read-window", str(nc), "u", "0", str(dec.y_offset), str(dec.x_offset),
                                  str(dec.ny_local), str(dec.nx_local), str(out)], check=True, capture_output=True, text=True)
            got_ny, got_nx, rss_kib = (int(v) for v in res.stdout.split())
            assert (got_ny, got_nx) == (n, n)
            assert rss_kib * 1024 < 2 * dec.ny_local * dec.nx_local * 8, "the window read held more than its own tile"
            t = np.fromfile(out).reshape(dec.ny_local + 2, dec.nx_local + 2)
            os.remove(out)
            ring = np.ones(t.shape, bool)
            ring[1:-1, 1:-1] = False
            t[ring] = 0.0  # the tool marks the untouched ring with -7; a fresh Field's ghosts are zero
            return t
        vr.upload_tiles(tile)
        os.remove(nc)
        want = _oracle_full(ora, u0, bc, D, vx, vy, dt, steps)
        del u0
        vr.advance(D, dt, vx, vy, steps)
        _compare_tiles(vr, want)
    finally:
        vr.close()
