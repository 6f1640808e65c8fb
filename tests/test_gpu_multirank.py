"""N > 1 on ONE GPU: 2/3/4-rank runs of the HIP stepper, one process per rank, all on device
0, halos carried by gloo through csim_stepper_halo_pack/_unpack (RCCL refuses two ranks on one
device; the RCCL calls themselves are covered by tests/test_gpu_comm.py).  Checked bit-for-bit
against the golden vectors of the reference's own `mpirun -np N` runs.  (Full-size decomposed runs and the
4 x 2 / 3 x 2 process grids: tests/test_gpu_virtual8.py, all ranks in one process, against the oracle.)"""
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


def _visible_gpus():
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.lib()
    return pkg.device_count()


@pytest.mark.parametrize("engine", ["hip-rccl", "hip-rccl3", "hip-rccl4", "hip-rccl1", "hip-rccl0"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_between_distinct_gpus_matches_the_reference_goldens(world, engine):
    """The one thing a one-GPU box cannot run: RCCL send/recv between DISTINCT GPUs (xGMI), every exchange
    schedule, checked against the goldens of the reference's own `mpirun -np N` runs.  Skipped unless at
    least `world` GPUs are visible — on the builder's one-GPU lease this has never executed."""
    if _visible_gpus() < world:
        pytest.skip(f"needs {world} visible GPUs")
    for case in cases_with(world):
        rc, out = launch(world, engine, case, timeout=900)
        assert rc == 0 and "ok=True" in out, (os.path.basename(case), out[-3000:])
