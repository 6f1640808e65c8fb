"""bench.py end to end on one GPU: the JSON contract of the N = 1 line (small grid, no CPU baseline)
and — through the self-linked torus test mode — the whole N > 1 code path: RCCL communicator, deep
faces in 8 directions, the trial of the exchange schedules, mass conservation across the seams."""
import json
import os
import subprocess
import sys

import pytest

from ports import rendezvous_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env=None, expect_rc=0):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "1536", "--ny", "1024", "--steps", "37",
           "--warmup", "7", "--ramp-seconds", "0.02", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert p.returncode == expect_rc, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # ONE JSON line and nothing else on stdout (RCCL / gloo banners belong on stderr)
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == lines, p.stdout[:2000]
    return json.loads(lines[0])


def test_bench_line_contract_single_gpu():
    r = run_bench([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 37 and r["warmup"] == 7 and r["dtype"] == "f64"
    assert r["unit"] == "Mcell-updates/s" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert abs(r["value"] - 1536 * 1024 * 37 / (r["ms_per_step"] * 37 * 1e-3) / 1e6) < 1e-6 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] <= 1.0
    assert rf["step_equivalent_x_peak"] == rf["step_equivalent_gbs"] / 8000.0
    assert r["roofline_valu"]["bound"] == "fp64-valu" and 0 < r["roofline_valu"]["frac"] < 1
    # one object prices HBM, the other the VALUs; exactly one of them is flagged as what binds the kernel
    assert rf["is_binding"] != r["roofline_valu"]["is_binding"] and "stored PMC profile" in rf["traffic_is"]
    cfg = r["config"]
    assert cfg["relative_mass_drift"] < 1e-9
    # median of >= 3 repeats, every sample in the line
    assert cfg["repeats"] == 3 and len(cfg["repeats_ms_per_step"]) == 3
    assert r["ms_per_step"] == sorted(cfg["repeats_ms_per_step"])[1]
    # the run checked itself: 40 steps of the bench field against the oracle's checksum of the same field
    pre = cfg["parity_preflight"]
    assert pre["ok"] and pre["checksum_expected"] and not pre.get("fixture_mismatch")
    (rec,) = pre["schedules"].values()
    assert rec["ok"] and rec["checksum_ok"] is True and rec["checksum"] == pre["checksum_expected"]
    assert cfg["stalled_schedule"] is None


def test_bench_configs1_diffusion_only_periodic_checks_itself_and_prices_hbm():
    """BASELINE configs[1] through bench.py (--physics 1.0,0.1,0,0 --bc pppp on 4096^2): the sweep's seven-operation
    flavour; the preflight compares with the ORACLE's checksum of that very workload (tests/golden/bench_checksum.json),
    and the line flags HBM, not the VALUs, as the binding resource"""
    r = run_bench(["--nx", "4096", "--ny", "4096", "--bc", "pppp", "--physics", "1.0,0.1,0,0", "--steps", "63"])
    cfg, rf = r["config"], r["roofline"]
    assert "D=1.0" in cfg["workload"] and "v=(0.0,0.0)" in cfg["workload"] and "bc=pppp" in cfg["workload"]
    pre = cfg["parity_preflight"]
    assert pre["ok"] and pre["checksum_expected"] and not pre.get("fixture_mismatch")
    (rec,) = pre["schedules"].values()
    assert rec["checksum_ok"] is True and rec["checksum"] == pre["checksum_expected"]
    assert rf["is_binding"] is True and r["roofline_valu"]["is_binding"] is False
    assert "advection term of zero velocity left out" in rf["kernel"] and rf["time_steps_per_launch"] == 7
    assert r["roofline_valu"]["useful_ops_per_cell_update"] == 8


def test_bench_multi_rank_path_on_self_linked_torus():
    r = run_bench([], env={"CSIM_BENCH_SELF_TORUS": "1"})
    cfg = r["config"]
    assert cfg["halo_transport"] == "rccl"
    sched = cfg["exchange_schedules_ms_per_step"]
    # overlap 0 (conservative, first), 5 (default: bulk-first with the stream relay), 1, 3 (merged launch), + "chosen"
    assert sched["chosen"] in sched and len(sched) == 5 and all(any(k.startswith(f"overlap-{m}") for k in sched)
                                                                for m in (0, 1, 3, 5))
    pr = cfg["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["kernel_avg_ms"] > 0 and pr[0]["neighbours"] == [0, 0, 0, 0]
    assert cfg["relative_mass_drift"] < 1e-9  # a lost or misplaced face would leak mass at the seams
    # parity preflight: every golden input through the RCCL path under every schedule that was then timed, plus the
    # 40-step checksum of the bench field, all schedules agreeing (test mode: with schedule 0, the torus has no golden)
    pre = cfg["parity_preflight"]
    assert pre["ok"] and len(pre["golden_cases"]) >= 10 and len(pre["schedules"]) == 4
    sums = {rec["checksum"] for rec in pre["schedules"].values()}
    assert len(sums) == 1 and all(rec["ok"] and rec["golden_ok"] for rec in pre["schedules"].values())
    assert cfg["stalled_schedule"] is None and cfg["repeats"] == 3 and len(cfg["repeats_ms_per_step"]) == 3
    # the conservative schedule is checked and timed first
    assert list(sched)[0].startswith("overlap-0") and list(pre["schedules"])[0].startswith("overlap-0")


def test_bench_stalled_schedule_still_yields_a_line():
    """a schedule that never comes back (test knob: schedule 3 hangs when its turn comes) must not cost the line: the
    watchdog prints it from the schedules already timed — the conservative one first — and the process leaves with
    status 3"""
    r = run_bench([], env={"CSIM_BENCH_SELF_TORUS": "1", "CSIM_BENCH_INJECT_STALL": "3", "CSIM_BENCH_PHASE_TIMEOUT": "3", "CSIM_BENCH_DEADLINE_SCALE": "0.1"},
                  expect_rc=3)
    cfg = r["config"]
    assert r["value"] > 0 and r["n_gpus"] == 1
    assert "overlap-3" in cfg["stalled_schedule"]["phase"] and "best COMPLETED" in cfg["value_is"]
    timed = {k: v for k, v in cfg["exchange_schedules_ms_per_step"].items() if isinstance(v, float)}
    assert any(k.startswith("overlap-0") for k in timed) and not any(k.startswith("overlap-3") for k in timed)
    assert abs(r["ms_per_step"] - min(timed.values())) < 1e-12
    assert cfg["parity_preflight"]["ok"]


def test_bench_checksum_fixture_is_what_the_oracle_computes():
    """tests/golden/bench_checksum.json, the value bench.py's preflight compares with at every N: re-derived here for
    the 16384^2 default workload — device-made hotspot downloaded, 40 steps by the ORACLE, numpy checksum — and
    compared with the HIP path's own checksum after the same steps."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    from oracle import cpu_oracle as ora
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_checksum.json")))["entries"]
    e = next(x for x in fx if (x["nx"], x["ny"], x["bc"]) == (16384, 16384, "dddd"))
    st = csim.Stepper.single(e["nx"], e["ny"], 1.0, 1.0, csim.bc_codes(e["bc"]))
    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
    ic = st.download()
    assert st.checksum() == e["checksum_ic"] == csim.checksum_host(ic[1:-1, 1:-1])
    for n in (1, 7, 32):
        st.run(e["D"], e["dt"], e["vx"], e["vy"], n)
    assert st.checksum() == e["checksum"]
    st.close()
    w = ora.World(16, e["nx"], e["ny"])
    w.scatter(np.ascontiguousarray(ic[1:-1, 1:-1]))
    del ic
    w.run(e["D"], e["vx"], e["vy"], e["dt"], ora.bc_codes(e["bc"]), e["steps"], threads=16)
    assert csim.checksum_host(w.gather()) == e["checksum"]


@pytest.mark.parametrize("forced", [True, False])
def test_bench_two_ranks_host_staged_fallback(forced):
    """two bench ranks sharing this GPU: the fall-back transport bench.py uses when the RCCL communicator cannot be
    built.  forced: CSIM_BENCH_HALO=gloo asks for it; not forced: the run first times its host-staged SAFETY-NET region,
    then really tries RCCL, which refuses two ranks on one device (ncclCommInitRank: invalid usage), and every rank
    falls back together"""
    port = str(rendezvous_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="1", CSIM_BENCH_PHASE_TIMEOUT="30")
        if forced:
            env.update(CSIM_BENCH_HALO="gloo", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "1536",
                                       "--ny", "1024", "--steps", "37", "--warmup", "7", "--ramp-seconds", "0.02"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    assert outs[0][0].strip() == lines[0] and outs[1][0].strip() == ""   # nothing but the one line on any rank's stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["halo_transport"].startswith("gloo")
    assert forced or "RCCL unavailable" in r["config"]["halo_transport"]
    assert r["config"]["relative_mass_drift"] < 1e-9
    # the parity preflight of a REAL two-rank run: every rank compared its tile of the reference's own `mpirun -np 2`
    # golden cases (ghost lines included) and the two ranks' checksums of the bench field add up to the oracle's value
    pre = r["config"]["parity_preflight"]
    assert pre["ok"] and len(pre["golden_cases"]) >= 7 and "mpirun -np 2" in pre["golden_reference"]
    (rec,) = pre["schedules"].values()
    assert rec["golden_ok"] is True and rec["checksum_ok"] is True and rec["checksum"] == pre["checksum_expected"]
