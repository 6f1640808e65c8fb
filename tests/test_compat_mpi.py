"""The reference's two MPI unit tests against the drop-in headers, under real MPI (no GPU needed: host code only).

driver/test_compat_mpi.cpp restates tests/simulation/unit/test_halo.cpp:8-66 (8 x 8 global grid, interior := rank id,
after exchange_halos every ghost line equals the neighbour's id) and test_decomp_mpi.cpp:6-35 against
include/climate/{decomp,field,halo}.hpp in the -DCSIM_WITH_MPI build — the MPI branch of driver/compat.cpp's
exchange_halos on host Fields — and compares Decomp2D::init with MPI's own MPI_Dims_create / MPI_Cart_* on the spot.
The reference runs them with `mpirun -np 4` (tests/simulation/CMakeLists.txt); here also on 2, 6 and 8 ranks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "driver")
MPIRUN = "/opt/conda/bin/mpirun"


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(MPIRUN):
        pytest.skip("no mpirun in this image")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", DRV, "test_compat_mpi"], check=True)
    return os.path.join(DRV, "test_compat_mpi")


@pytest.mark.parametrize("ranks", [4, 2, 6, 8, 1])
def test_reference_mpi_unit_tests_against_the_drop_in_headers(exe, ranks):
    r = subprocess.run([MPIRUN, "-np", str(ranks), exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and f"{ranks} ranks, all passed" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
