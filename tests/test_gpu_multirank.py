"""N > 1 on ONE GPU: 2/3/4-rank runs of the HIP stepper, one process per rank, all on device
0, halos carried by gloo through csim_stepper_halo_pack/_unpack (RCCL refuses two ranks on one
device; the RCCL calls themselves are covered by tests/test_gpu_comm.py).  Checked bit-for-bit
against the golden vectors of the reference's own `mpirun -np N` runs."""
import os

import pytest

from test_multirank_gloo import cases_with, launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3, 4])  # this process + 4 ranks stays under the box's cap
def test_hip_stepper_multirank_one_gpu(world):
    for case in cases_with(world):
        rc, out = launch(world, "hip-external", case, timeout=600)
        assert rc == 0 and "ok=True" in out, (os.path.basename(case), out[-3000:])


@pytest.mark.parametrize("depth", [2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_hip_stepper_fused_passes_multirank_one_gpu(world, depth):
    """2..7 steps per pass across ranks: deep faces incl. the diagonal corner blocks"""
    # depths 6 and 7 on EVERY golden case of this world size (odd widths, tiny tiles, all BC mixes, the
    # three division modes); the other depths on the wide cases
    cases = cases_with(world) if depth in (6, 7) else [c for c in cases_with(world) if "run_fused" in c]
    assert cases
    for case in cases:
        rc, out = launch(world, f"hip-external{depth}", case, timeout=600)
        assert rc == 0 and "ok=True" in out, (os.path.basename(case), out[-3000:])
