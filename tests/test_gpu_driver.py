"""The C++17 host side on a GPU box: (1) driver/test_compat — the reference's unit-test bodies
against the source-compatible headers include/climate/*.hpp; (2) the `climate_sim_hip` driver
end to end (config -> IC -> GPU time loop -> CDF-5 snapshots), its snapshot records compared
bit-for-bit with the golden vectors; (3) where mpirun exists, the MPI-launched flavour with 4
ranks sharing the GPU and reference-style MPI faces (--halo=mpi)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from test_host_config_snapshot import parse_cdf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "driver")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MPIRUN = "/opt/conda/bin/mpirun"


def test_compat_headers_unit_tests():
    r = subprocess.run([os.path.join(DRV, "test_compat")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all passed" in r.stdout, r.stdout + r.stderr


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def run_driver(tmp_path, exe, m, steps_arg, out_every, launcher=(), extra=()):
    bc = {"d": "dirichlet", "n": "neumann", "p": "periodic"}
    args = [f"--nx={m['nx']}", f"--ny={m['ny']}", f"--dx={m['dx']}", f"--dy={m['dy']}", f"--D={m['D']}",
            f"--vx={m['vx']}", f"--vy={m['vy']}", f"--dt={m['dt']}", f"--steps={steps_arg}",
            f"--out_every={out_every}", f"--bc.left={bc[m['bc'][0]]}", f"--bc.right={bc[m['bc'][1]]}",
            f"--bc.bottom={bc[m['bc'][2]]}", f"--bc.top={bc[m['bc'][3]]}",
            f"--ic.sigma_frac={m['sigma_frac']}"]
    r = subprocess.run([*launcher, os.path.join(DRV, exe), *args, *extra], capture_output=True, text=True,
                       cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout, parse_cdf(os.path.join(tmp_path, "outputs", "snapshots.nc"))


def records(h, m):
    v = h["vars"]["u"]
    n = h["numrecs"]
    return np.frombuffer(h["raw"], dtype=">f8", count=n * m["ny"] * m["nx"], offset=v["begin"]) \
        .reshape(n, m["ny"], m["nx"])


@pytest.mark.parametrize("case", ["run_dev_yaml_small", "run_neumann_negv_gauss", "run_dt_clamped"])
def test_driver_single_rank_snapshots_match_golden(tmp_path, case):
    z, m = golden(case)
    out, h = run_driver(tmp_path, "climate_sim_hip", m, m["steps"] + 1, m["steps"])
    assert re.search(r"timing: total_max=[0-9.e+-]+ s, worst_avg_step=[0-9.e+-]+ s", out)
    assert "IC min/max: 0 / " in out  # min over the array incl. zero ghosts (reference main.cpp:73-77)
    rec = records(h, m)
    assert rec.shape[0] == 2            # states before step 0 and before step `steps`
    assert np.array_equal(rec[0], z["u0"])
    assert np.array_equal(rec[1], z["u_final"])
    if case == "run_dt_clamped":
        assert float(z["dt_effective"]) < m["dt"]


@pytest.mark.skipif(not os.path.exists(MPIRUN), reason="no mpirun in this image")
def test_driver_mpi_four_ranks_one_gpu(tmp_path):
    exe = os.path.join(DRV, "climate_sim_hip_mpi")
    if not os.path.exists(exe):
        pytest.skip("MPI flavour not built")
    z, m = golden("run_dev_yaml_small")
    out, h = run_driver(tmp_path, "climate_sim_hip_mpi", m, m["steps"] + 1, m["steps"],
                        launcher=(MPIRUN, "-np", "4"), extra=("--halo=mpi", "--checksum"))
    rec = records(h, m)
    assert np.array_equal(rec[0], z["u0"]) and np.array_equal(rec[1], z["u_final"])
    assert "dims=2x2" in out
    # --checksum: the four ranks' position-weighted checksums add up (MPI_SUM, mod 2^64) to the value of the global field
    import re
    from __graft_entry__ import load_package
    got = int(re.search(r"checksum: (0x[0-9a-f]{16})", out).group(1), 16)
    # (the driver ran steps + 1 steps so that the record of step `steps` exists: the oracle takes the one more step)
    from oracle import cpu_oracle as ora
    last = np.zeros((m["ny"] + 2, m["nx"] + 2))
    last[1:-1, 1:-1] = z["u_final"]
    ora.run_single(last, m["dx"], m["dy"], m["D"], m["vx"], m["vy"], float(z["dt_effective"]), ora.bc_codes(m["bc"]), 1)
    assert got == load_package().checksum_host(last[1:-1, 1:-1])


def test_driver_ic_from_netcdf_file(tmp_path):
    """BASELINE config 5's "NetCDF initial-condition load": the reference has no reader (ic.mode=file
    throws there); ours loads a classic NetCDF (y,x) double variable.  A random field
    written with the snapshot writer must step exactly like the golden run that started from it."""
    z, m = golden("run_fused_256x48")
    raw = tmp_path / "ic.bin"
    np.ascontiguousarray(z["u0"]).tofile(raw)
    ic = tmp_path / "ic.nc"
    subprocess.run([os.path.join(DRV, "csim_hosttool"), "nc-write", str(ic), str(raw), "1",
                    f"--nx={m['nx']}", f"--ny={m['ny']}"], check=True)
    m2 = dict(m, sigma_frac=0.05)
    out, h = run_driver(tmp_path, "climate_sim_hip", m2, m["steps"] + 1, m["steps"],
                        extra=("--ic.mode=file", f"--ic.path={ic}"))
    rec = records(h, m)
    assert np.array_equal(rec[0], z["u0"]) and np.array_equal(rec[1], z["u_final"])


@pytest.mark.skipif(not os.path.exists(MPIRUN), reason="no mpirun in this image")
@pytest.mark.parametrize("case,np_ranks,dims", [("run_fused_256x48", 4, "2x2"), ("run_mixed_bc_random", 2, "2x1")])
def test_driver_mpi_ranks_window_the_netcdf_ic(tmp_path, case, np_ranks, dims):
    """config 5's shape at test size: a decomposed run whose ranks each read ONLY their block of the IC
    file (read_netcdf_window: per-rank start/count like reference src/io.cpp:402-418), then step with
    MPI faces; the snapshot records must equal the golden run from the same initial field."""
    exe = os.path.join(DRV, "climate_sim_hip_mpi")
    if not os.path.exists(exe):
        pytest.skip("MPI flavour not built")
    z, m = golden(case)
    raw = tmp_path / "ic.bin"
    np.ascontiguousarray(z["u0"]).tofile(raw)
    ic = tmp_path / "ic.nc"
    subprocess.run([os.path.join(DRV, "csim_hosttool"), "nc-write", str(ic), str(raw), "1",
                    f"--nx={m['nx']}", f"--ny={m['ny']}"], check=True)
    m2 = dict(m, sigma_frac=0.05)
    out, h = run_driver(tmp_path, "climate_sim_hip_mpi", m2, m["steps"] + 1, m["steps"],
                        launcher=(MPIRUN, "-np", str(np_ranks)),
                        extra=("--halo=mpi", "--ic.mode=file", f"--ic.path={ic}"))
    rec = records(h, m)
    assert np.array_equal(rec[0], z["u0"]) and np.array_equal(rec[1], z["u_final"])
    assert f"dims={dims}" in out


@pytest.mark.skipif(not os.path.exists(MPIRUN), reason="no mpirun in this image")
def test_driver_config5_shape_at_4101x4099_four_mpi_ranks(tmp_path):
    """BASELINE configs[4] end to end at a size the oracle finishes in seconds: all-Neumann, the IC loaded
    from a NetCDF file by four MPI ranks that each read only their block (remainder tiles: 2050 + 2051 columns,
    2049 + 2050 rows), faces over MPI once per fused pass, 24 steps; the snapshot record of the state before
    step 24 must equal the oracle's 4-tile run bit for bit."""
    from oracle import cpu_oracle as ora
    exe = os.path.join(DRV, "climate_sim_hip_mpi")
    if not os.path.exists(exe):
        pytest.skip("MPI flavour not built")
    nx, ny, steps = 4101, 4099, 24
    u0 = ora.gaussian_global(nx, ny, sigma_frac=0.07)[1:-1, 1:-1] + 0.05 * np.random.default_rng(5).random((ny, nx))
    raw = tmp_path / "ic.bin"
    np.ascontiguousarray(u0).tofile(raw)
    ic = tmp_path / "ic.nc"
    subprocess.run([os.path.join(DRV, "csim_hosttool"), "nc-write", str(ic), str(raw), "1", f"--nx={nx}", f"--ny={ny}"],
                   check=True)
    m = dict(nx=nx, ny=ny, dx=1.0, dy=1.0, D=0.05, vx=0.5, vy=0.25, dt=0.1, bc="nnnn", sigma_frac=0.05)
    out, h = run_driver(tmp_path, "climate_sim_hip_mpi", m, steps + 1, steps, launcher=(MPIRUN, "-np", "4"),
                        extra=("--halo=mpi", "--ic.mode=file", f"--ic.path={ic}"))
    assert "dims=2x2" in out
    rec = records(h, m)
    assert rec.shape[0] == 2 and np.array_equal(rec[0], u0)
    w = ora.World(4, nx, ny)
    w.scatter(np.ascontiguousarray(u0))
    w.run(m["D"], m["vx"], m["vy"], m["dt"], ora.bc_codes("nnnn"), steps, threads=4)
    assert np.array_equal(rec[1], w.gather())


def test_driver_rejects_bad_ic(tmp_path):
    """reference tests/simulation/integration/integration_boundary_error.cpp: a bad IC preset gives
    a non-zero exit and no output file."""
    r = subprocess.run([os.path.join(DRV, "climate_sim_hip"), "--nx=32", "--ny=32", "--steps=2",
                        "--ic.preset=no_such_preset"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0
    assert not os.path.exists(os.path.join(tmp_path, "outputs", "snapshots.nc"))
