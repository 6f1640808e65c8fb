"""the benchmark / scaling harness.  tools/scaling_report.py and tools/run_benchmark.sh
restate the reference's scripts/run_benchmark.sh (strong + weak scaling CSVs, `timing: total_max=` regex
contract :37, speedup / efficiency / Karp-Flatt :54-68).  No GPU: the shell script is run against a
stand-in driver that prints the driver's timing line for a modelled run time."""
import csv
import io
import json
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def karp_flatt(s, p):
    return (1.0 / s - 1.0 / p) / (1.0 - 1.0 / p)


def test_scaling_report_arithmetic(tmp_path):
    import scaling_report
    ms = {1: 0.184, 2: 0.095, 4: 0.050, 8: 0.0275}
    files = []
    for n, t in ms.items():
        line = dict(n_gpus=n, ms_per_step=t, value=16384 * 16384 / t / 1e3,
                    config=dict(hbm_gbs_whole_job=4200.0 * n), roofline=dict(frac=0.53))
        f = tmp_path / f"n{n}.json"
        f.write_text("some log noise\n" + json.dumps(line) + "\n")
        files.append(str(f))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_report.py"), *reversed(files)],
                         capture_output=True, text=True, check=True).stdout
    rows = list(csv.DictReader(io.StringIO(out)))
    assert [int(r["n_gpus"]) for r in rows] == [1, 2, 4, 8]
    for r in rows:
        n = int(r["n_gpus"])
        s = ms[1] / ms[n]
        assert abs(float(r["speedup"]) - s) < 1e-3
        assert abs(float(r["efficiency"]) - s / n) < 1e-3
        assert abs(float(r["mcell_updates_per_s"]) - 16384 * 16384 / ms[n] / 1e3) < 1.0
        assert abs(float(r["step_equivalent_gb_per_s"]) - 16384 * 16384 * 16 / ms[n] / 1e6) < 1.0
        assert float(r["hbm_gb_per_s_measured"]) == 4200.0 * n and float(r["kernel_hbm_roofline_frac"]) == 0.53
        if n == 1:
            assert r["karp_flatt"] == ""
        else:
            assert abs(float(r["karp_flatt"]) - karp_flatt(s, n)) < 1e-5
    # N = 1 missing: the smallest count stands in for it (extrapolated 1-GPU time)
    rows2 = scaling_report.table([dict(n_gpus=2, ms_per_step=0.1, value=1.0), dict(n_gpus=4, ms_per_step=0.06, value=1.0)])
    assert rows2[1].split(",")[6] == "2.0000" and rows2[2].split(",")[6] == f"{0.2 / 0.06:.4f}"


FAKE_DRIVER = """#!/usr/bin/env bash
# stand-in for climate_sim_hip[_mpi]: run time = cells * steps / (1e9 cell-updates/s per rank * P) + 1 ms per rank of "serial" cost
P=${FAKE_P:-1}
for a in "$@"; do case $a in --nx=*) NX=${a#--nx=};; --ny=*) NY=${a#--ny=};; --steps=*) STEPS=${a#--steps=};; esac; done
T=$(python3 -c "print($NX*$NY*$STEPS/(1e9*$P) + 0.001*$P)")
echo "climate-sim-mpi-cpp "
echo "timing: total_max=$T s, worst_avg_step=0.001 s"
"""
FAKE_MPIRUN = """#!/usr/bin/env bash
# stand-in for mpirun -np P exe args...
[ "$1" = "-np" ] || exit 9
export FAKE_P=$2; shift 2
exec "$@"
"""


def _fake_tools(tmp_path):
    drv = tmp_path / "drv"
    drv.mkdir()
    for name in ("climate_sim_hip", "climate_sim_hip_mpi"):
        f = drv / name
        f.write_text(FAKE_DRIVER)
        f.chmod(f.stat().st_mode | stat.S_IEXEC)
    m = tmp_path / "mpirun"
    m.write_text(FAKE_MPIRUN)
    m.chmod(m.stat().st_mode | stat.S_IEXEC)
    return dict(os.environ, DRV=str(drv), MPIRUN=str(m), OUT=str(tmp_path / "results"))


def _model(nx, ny, steps, p):
    return nx * ny * steps / (1e9 * p) + 0.001 * p


def test_run_benchmark_strong_csv(tmp_path):
    env = _fake_tools(tmp_path)
    env.update(NX="2048", NY="1024", STEPS="50")
    subprocess.run(["bash", os.path.join(ROOT, "tools", "run_benchmark.sh"), "strong", "1", "2", "4", "8"],
                   env=env, check=True, capture_output=True, text=True)
    rows = list(csv.DictReader(open(tmp_path / "results" / "strong.csv")))
    assert [int(r["ranks"]) for r in rows] == [1, 2, 4, 8]
    t1 = _model(2048, 1024, 50, 1)
    for r in rows:
        p = int(r["ranks"])
        t = _model(2048, 1024, 50, p)
        assert (int(r["nx"]), int(r["ny"]), int(r["steps"])) == (2048, 1024, 50)
        assert abs(float(r["total_max_s"]) - t) < 1e-6
        assert abs(float(r["mcell_updates_per_s"]) - 2048 * 1024 * 50 / t / 1e6) < 0.1
        s = t1 / t
        assert abs(float(r["speedup"]) - s) < 1e-3 and abs(float(r["efficiency"]) - s / p) < 1e-3
        assert r["karp_flatt"] == "" if p == 1 else abs(float(r["karp_flatt"]) - karp_flatt(s, p)) < 1e-5


def test_run_benchmark_weak_csv(tmp_path):
    """weak mode: a fixed TILE x TILE per rank, ranks laid out like MPI_Dims_create (2 -> 2x1, 4 -> 2x2,
    8 -> 4x2), scaled speedup P * T1 / Tp (reference scripts/run_benchmark.sh weak section)"""
    env = _fake_tools(tmp_path)
    env.update(TILE="512", STEPS="20")
    subprocess.run(["bash", os.path.join(ROOT, "tools", "run_benchmark.sh"), "weak", "1", "2", "4", "8"],
                   env=env, check=True, capture_output=True, text=True)
    rows = list(csv.DictReader(open(tmp_path / "results" / "weak.csv")))
    grids = {1: (512, 512), 2: (1024, 512), 4: (1024, 1024), 8: (2048, 1024)}
    t1 = _model(512, 512, 20, 1)
    for r in rows:
        p = int(r["ranks"])
        assert (int(r["nx"]), int(r["ny"])) == grids[p]
        t = _model(*grids[p], 20, p)
        assert abs(float(r["total_max_s"]) - t) < 1e-6
        s = p * t1 / t
        assert abs(float(r["speedup"]) - s) < 1e-3 and abs(float(r["efficiency"]) - s / p) < 1e-3
