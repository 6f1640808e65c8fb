"""bench.py's host logic without a GPU: the line is built from the measurement records alone (`build_line`), so its
protocol arithmetic — median of the timed regions, the fall-back to the best completed schedule trial when a run
stalled, the roofline objects — and the watchdog (a deadline enforced from a thread, SIGTERM taken over from the
launcher) can be checked in the build container.  The measured path itself: tests/test_gpu_bench.py."""
import importlib.util
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _state(bench, steps=20, world=1, local=(16384, 16384)):
    args = types.SimpleNamespace(steps=steps, warmup=5, nx=16384, ny=16384, bc="dddd", contract=0, no_overlap=False,
                                 no_preflight=False, rows_per_chunk=0)
    dec = types.SimpleNamespace(nx_local=local[0], ny_local=local[1], dims=[4, 2] if world == 8 else [1, 1])
    return dict(args=args, rank=0, world=world, self_torus=False, multi=world > 1, cpu=None, dec=dec, dt=0.1,
                halo="rccl" if world > 1 else "none", measurements={}, chosen=None, final=[], preflight=dict(ok=True, _first_checksum=7),
                stalled=None, exchange_modes=None, mass_drift=1e-16, minmax=(0.0, 1.0), ramp_steps=1000, tuned_rows=182)


def _region(ms_per_step, steps, T=7, schedule=None, launches=2):
    return dict(schedule=schedule, elapsed=ms_per_step * steps * 1e-3, elapsed_local=ms_per_step * steps * 1e-3, T=T,
                kern_ms=1.2 * launches, launches=launches, comm_ms=0.0, comm_n=0, last_rows=182, overlap=5)


def test_line_is_the_median_region_and_carries_every_sample(bench):
    S = _state(bench)
    S["final"] = [_region(0.172, 20), _region(0.170, 20), _region(0.175, 20)]
    line = bench.build_line(S)
    assert line["ms_per_step"] == pytest.approx(0.172) and line["steps"] == 20 and line["n_gpus"] == 1
    assert line["value"] == pytest.approx(16384 * 16384 * 20 / (0.172e-3 * 20) / 1e6)
    cfg = line["config"]
    assert cfg["repeats"] == 3 and cfg["repeats_ms_per_step"] == pytest.approx([0.172, 0.170, 0.175])
    assert cfg["stalled_schedule"] is None and "median" in cfg["value_is"]
    assert "_first_checksum" not in cfg["parity_preflight"] and cfg["parity_preflight"]["ok"] is True
    rf, rv = line["roofline"], line["roofline_valu"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["frac"] == pytest.approx(rf["achieved"] / 8000.0)
    assert rf["kernel_avg_ms"] == pytest.approx(1.2) and rf["launches_timed"] == 6      # all regions' brackets together
    assert rf["is_binding"] is False and rv["is_binding"] is True and "stored PMC profile" in rf["traffic_is"]
    assert rf["traffic"] and 1.0 < rf["traffic_over_algorithmic"] < 1.2                  # profiles/pmc_traffic.json, T = 7
    json.dumps(line)                                                                    # serialisable as it stands


def test_stalled_run_falls_back_to_the_best_completed_trial(bench):
    S = _state(bench, world=8, local=(4096, 8192))
    S["measurements"] = {0: [_region(0.034, 20, schedule=0)], 5: [_region(0.029, 20, schedule=5)]}
    S["exchange_modes"] = {"overlap-0": 0.034, "overlap-5": 0.029}
    S["stalled"] = dict(phase="parity preflight of overlap-3", why="no progress")
    line = bench.build_line(S)
    assert line["ms_per_step"] == pytest.approx(0.029) and line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["exchange_schedule"] == 5 and "best COMPLETED" in line["config"]["value_is"]
    assert line["config"]["stalled_schedule"]["phase"].endswith("overlap-3")
    S["measurements"] = {}
    assert bench.build_line(S) is None                                                   # nothing timed yet: no line


def test_preflight_inputs_exist_for_every_world_size_the_driver_runs(bench):
    for world, at_least in ((2, 7), (4, 8), (8, 5)):
        cases = bench.golden_cases(world)
        assert len(cases) >= at_least
        for name, z, m in cases:
            assert world in m["ranks"] and f"local_np{world}_rank{world - 1}" in z.files
    e = bench.expected_checksum(16384, 16384, "dddd", 0.1)
    assert e and e["steps"] == sum(bench.CHECK_STEPS) and e["checksum"] == int(e["checksum_hex"], 16)
    assert bench.expected_checksum(16384, 16384, "dddd", 0.05) is None and bench.expected_checksum(100, 100, "dddd", 0.1) is None


WATCHDOG_CHILD = """
import os, sys, time
sys.path.insert(0, {root!r})
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join({root!r}, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
def emit(phase, why):
    os.write(1, ("LINE phase=%s why=%s\\n" % (phase, why)).encode())
wd = bench.Watchdog(0, emit)
wd.arm({deadline}, "the phase under test")
print("armed", flush=True)
time.sleep(60)          # stands for a native call that never returns
print("NOT REACHED", flush=True)
"""


def test_watchdog_prints_the_line_and_leaves_with_status_3_on_a_stall():
    p = subprocess.run([sys.executable, "-c", WATCHDOG_CHILD.format(root=ROOT, deadline=0.5)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "LINE phase=the phase under test why=no progress" in p.stdout and "NOT REACHED" not in p.stdout


def test_watchdog_takes_sigterm_from_the_launcher():
    import signal
    import time
    p = subprocess.Popen([sys.executable, "-c", WATCHDOG_CHILD.format(root=ROOT, deadline=1000)], stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "armed"
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)          # what torch.distributed.run sends to the surviving ranks
    out = p.communicate(timeout=30)[0]
    assert p.returncode == 3 and "LINE phase=the phase under test why=SIGTERM" in out
