"""bench.py's multi-rank CONTROL FLOW on CPUs: the real main() with a mock engine (tests/bench_mock_runner.py), one
process per rank over torch.distributed gloo.  What runs here for the first time outside a multi-GPU node: every rank
joining the same collectives in the same order through safety net, communicator, per-schedule parity preflight (the
reference's golden cases for that world size), schedule trials and the three final regions; a schedule that computes
wrong tiles being excluded; a rank that hangs under one schedule — the other ranks then sit in a collective — ending
with the watchdogs' line and status 3; the same under the very first schedule ending with the safety-net region."""
import json
import os
import subprocess
import sys

import pytest

from ports import rendezvous_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNER = os.path.join(ROOT, "tests", "bench_mock_runner.py")


def run(world, env=None, timeout=300):
    port = str(rendezvous_port())
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                 OMP_NUM_THREADS="1", **(env or {}))
        procs.append(subprocess.Popen([sys.executable, RUNNER, "--gpus", str(world), "--steps", "20", "--warmup", "5",
                                       "--ramp-seconds", "0.01", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    outs = [p.communicate(timeout=timeout) for p in procs]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    for r in range(1, world):
        assert not [ln for ln in outs[r][0].splitlines() if ln.strip()], f"rank {r} wrote to stdout"
    return [p.returncode for p in procs], (json.loads(lines[0]) if len(lines) == 1 else None), outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_complete_run_every_rank_joins_every_collective(world):
    rcs, line, outs = run(world)
    assert rcs == [0] * world and line is not None, outs[0][1][-2000:]
    cfg = line["config"]
    assert line["n_gpus"] == world and line["scaling"] == "strong" and cfg["halo_transport"] == "rccl"
    dims = {2: "2x1", 4: "2x2", 8: "4x2"}[world]
    assert f"decomp {dims}" in cfg["workload"] and len(cfg["per_rank"]) == world
    assert [p["rank"] for p in cfg["per_rank"]] == list(range(world))
    pre = cfg["parity_preflight"]
    assert pre["ok"] and f"mpirun -np {world}" in pre["golden_reference"] and len(pre["golden_cases"]) >= 5
    assert len(pre["schedules"]) == 4 and all(r["golden_ok"] and r["checksum_ok"] for r in pre["schedules"].values())
    sched = cfg["exchange_schedules_ms_per_step"]
    assert list(sched)[0].startswith("overlap-0") and any(k.startswith("safety net") for k in sched) and sched["chosen"] in sched
    assert cfg["repeats"] == 3 and cfg["stalled_schedule"] is None and "median" in cfg["value_is"]


def test_a_schedule_that_computes_wrong_tiles_is_excluded_not_timed():
    rcs, line, outs = run(4, env={"MOCK_BAD_SCHEDULE": "1"})
    assert rcs == [0] * 4 and line is not None
    cfg = line["config"]
    bad = [k for k in cfg["exchange_schedules_ms_per_step"] if k.startswith("overlap-1")][0]
    assert cfg["exchange_schedules_ms_per_step"][bad].startswith("EXCLUDED")
    assert cfg["parity_preflight"]["ok"] is False and cfg["parity_preflight"]["schedules"][bad]["golden_ok"] is False
    assert cfg["exchange_schedule"] != 1 and "PARITY FAILURE" in outs[0][1]


def test_one_rank_hanging_under_a_later_schedule_still_yields_the_line():
    """rank 2 never returns from its first run() under schedule 3: ranks 0, 1, 3 then wait in a collective; every
    watchdog fires (rank 0 first), the line is built from the schedules already timed, all ranks leave with status 3"""
    rcs, line, outs = run(4, env={"MOCK_STALL_RANK": "2", "MOCK_STALL_SCHEDULE": "3", "CSIM_BENCH_PHASE_TIMEOUT": "3", "CSIM_BENCH_DEADLINE_SCALE": "0.1"})
    # (rank 0 leaves with the watchdog's status; the others either through their own watchdog or through gloo noticing that
    # rank 0 is gone: non-zero either way)
    assert rcs[0] == 3 and all(rcs) and line is not None, (rcs, [o[1][-500:] for o in outs])
    cfg = line["config"]
    assert "overlap-3" in cfg["stalled_schedule"]["phase"] and "best COMPLETED" in cfg["value_is"]
    timed = {k: v for k, v in cfg["exchange_schedules_ms_per_step"].items() if isinstance(v, float)}
    assert any(k.startswith("overlap-0") for k in timed) and any(k.startswith("overlap-5") for k in timed)
    assert line["value"] > 0 and cfg["halo_transport"] == "rccl"


def test_a_hang_in_the_very_first_exchange_leaves_the_safety_net_region():
    rcs, line, outs = run(2, env={"MOCK_STALL_RANK": "1", "MOCK_STALL_SCHEDULE": "0", "CSIM_BENCH_PHASE_TIMEOUT": "3", "CSIM_BENCH_DEADLINE_SCALE": "0.1"}, timeout=400)
    assert rcs[0] == 3 and all(rcs) and line is not None, (rcs, [o[1][-500:] for o in outs])
    cfg = line["config"]
    assert cfg["halo_transport"].startswith("gloo (host-staged): the SAFETY-NET region")
    assert "overlap-0" in cfg["stalled_schedule"]["phase"] and line["value"] > 0


def test_under_the_drivers_own_launcher_the_line_survives_a_stalled_rank(tmp_path):
    """the exact launch line of the driver (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...): rank 1 hangs under schedule 3, rank 0's watchdog prints the line and
    leaves, the launcher then terminates rank 1 (SIGTERM, taken by its watchdog) and reports failure — with the one JSON
    line on its stdout"""
    port = str(rendezvous_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", MOCK_STALL_RANK="1", MOCK_STALL_SCHEDULE="3", CSIM_BENCH_PHASE_TIMEOUT="3",
               CSIM_BENCH_DEADLINE_SCALE="0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, RUNNER, "--gpus", "2", "--steps", "20", "--warmup", "5", "--ramp-seconds", "0.01",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, env=env, cwd=tmp_path)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert p.returncode != 0 and len(lines) == 1, (p.returncode, p.stdout[-1000:], p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "overlap-3" in line["config"]["stalled_schedule"]["phase"]


def test_under_the_drivers_own_launcher_a_complete_run(tmp_path):
    port = str(rendezvous_port())
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", port, RUNNER, "--gpus", "4", "--steps", "20", "--warmup", "5", "--ramp-seconds", "0.01",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, env=dict(os.environ, OMP_NUM_THREADS="1"),
                       cwd=tmp_path)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert p.returncode == 0 and len(lines) == 1 and lines[0].startswith('{"metric"'), (p.returncode, p.stdout[-1000:], p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["config"]["parity_preflight"]["ok"] and line["config"]["stalled_schedule"] is None
