"""N > 1 on CPU: world_size 2/4 gloo runs of the hot path with the product's MPI-free
decomposition (csim_decomp_init) and the reference's halo message schedule, stepped by the
oracle, against the golden vectors the compiled reference produced under `mpirun -np N`.
(The same worker drives the HIP stepper on a GPU box: tests/test_gpu_multirank.py.)"""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
WORKER = os.path.join(HERE, "multirank_worker.py")


from ports import rendezvous_port as free_port  # noqa: E402  (below the ephemeral range: see tests/ports.py)


def launch(world, engine, case, timeout=300):
    """start `world` worker processes directly (env rendezvous on 127.0.0.1, no launcher process:
    a GPU box allows few processes on its card).  A run that ends WITHOUT a verdict line — a rank stuck in the
    rendezvous or elsewhere: the workers dump their stacks and leave after 150 s — is started once more on a fresh
    port, and the first attempt's output is raised as a warning so that it stays visible; a run that printed its
    verdict (ok=True / ok=False) is never repeated."""
    rc, out = _launch_once(world, engine, case, timeout)
    if rc != 0 and "MULTIRANK engine=" not in out:
        import warnings
        warnings.warn(f"multi-rank worker run without a verdict (rc {rc}), repeated once; first attempt said:\n{out[-4000:]}")
        rc, out = _launch_once(world, engine, case, timeout)
    return rc, out


def _launch_once(world, engine, case, timeout):
    port = str(free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, OMP_NUM_THREADS="1", RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, WORKER, "--engine", engine, "--case", case],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                      env=env))
    out, rc = "", 0
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            rc = rc or 124
        out += o
        rc = rc or p.returncode
    return rc, out


def cases_with(world):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz"))):
        m = json.loads(str(np.load(p, allow_pickle=False)["meta"]))
        if world in m["ranks"]:
            out.append(p)
    return out


@pytest.mark.parametrize("case", cases_with(2), ids=lambda p: os.path.basename(p)[:-4])
def test_world2_gloo_oracle(case):
    rc, out = launch(2, "oracle", case)
    assert rc == 0 and "ok=True" in out, out[-2000:]


@pytest.mark.parametrize("case", cases_with(4)[:3], ids=lambda p: os.path.basename(p)[:-4])
def test_world4_gloo_oracle(case):
    rc, out = launch(4, "oracle", case)
    assert rc == 0 and "ok=True" in out, out[-2000:]


@pytest.mark.parametrize("world", [6, 8])
def test_world6_and_8_gloo_oracle(world):
    """3x2 and 4x2 process grids (the 8-GPU decomposition of BASELINE configs[4]): ranks with both
    x-neighbours, a y-neighbour and two diagonal peers"""
    cases = cases_with(world)
    assert cases
    for case in cases:
        rc, out = launch(world, "oracle", case)
        assert rc == 0 and "ok=True" in out, out[-2000:]
