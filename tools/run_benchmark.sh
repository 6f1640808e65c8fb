#!/usr/bin/env bash
# tools/run_benchmark.sh — the reference's scaling benchmark (scripts/run_benchmark.sh: strong
# scaling on a fixed grid, weak scaling with a fixed tile per rank, parsing the driver's
# `timing: total_max=` line) for the MI355X driver.  One rank per GPU.
#   tools/run_benchmark.sh [strong|weak] [GPU counts...]      e.g.  tools/run_benchmark.sh strong 1 2 4 8
# Needs climate_sim_hip_mpi (make -C climate-sim-mpi-cpp_amd/driver mpi) and mpirun for counts > 1;
# count 1 uses climate_sim_hip.  Unlike the reference script, physics is switched on (its
# defaults run D = v = 0) and snapshots are off in the timed run (--no-output).
# DRV / MPIRUN / OUT can be overridden (tests/test_scaling_tools.py runs the script against a stand-in driver).
# Output: bench/results/<mode>.csv with ranks, grid, seconds, Mcell-updates/s, speedup,
# efficiency and the Karp-Flatt serial fraction.
set -euo pipefail
MODE=${1:-strong}; shift || true
COUNTS=("$@"); [ ${#COUNTS[@]} -eq 0 ] && COUNTS=(1)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DRV=${DRV:-$ROOT/climate-sim-mpi-cpp_amd/driver}
MPIRUN=${MPIRUN:-/opt/conda/bin/mpirun}
NX=${NX:-16384}; NY=${NY:-16384}; STEPS=${STEPS:-200}
TILE=${TILE:-8192}
PHYS="--D=0.05 --vx=0.5 --vy=0.25 --dt=0.1"
OUT=${OUT:-$ROOT/bench/results}; mkdir -p "$OUT"
CSV=$OUT/$MODE.csv
echo "ranks,nx,ny,steps,total_max_s,mcell_updates_per_s,speedup,efficiency,karp_flatt" > "$CSV"
T1=""
for P in "${COUNTS[@]}"; do
  if [ "$MODE" = weak ]; then   # fixed TILE x TILE per rank, ranks laid out like MPI_Dims_create
    read PX PY < <(python3 -c "
p=$P; b=max(f for f in range(1,int(p**0.5)+1) if p%f==0); print(p//b, b)")
    GX=$((TILE*PX)); GY=$((TILE*PY))
  else
    GX=$NX; GY=$NY
  fi
  ARGS="--nx=$GX --ny=$GY --steps=$STEPS $PHYS --no-output --device-ic"
  if [ "$P" -eq 1 ]; then LOG=$("$DRV/climate_sim_hip" $ARGS)
  else LOG=$("$MPIRUN" -np "$P" "$DRV/climate_sim_hip_mpi" $ARGS); fi
  T=$(echo "$LOG" | sed -n 's/.*timing: total_max=\([0-9.eE+-]*\) s.*/\1/p')
  [ -z "$T1" ] && T1=$(python3 -c "print($T*$P)")   # 1-rank time (extrapolated if the first count is not 1)
  python3 - "$P" "$GX" "$GY" "$STEPS" "$T" "$T1" "$MODE" >> "$CSV" <<'PY'
import sys
p, gx, gy, steps, t, t1, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6]), sys.argv[7]
mc = gx * gy * steps / t / 1e6
sp = (t1 / t) if mode == "strong" else (t1 / t) * p      # weak: scaled speedup
eff = sp / p
kf = "" if p == 1 else f"{(1/sp - 1/p) / (1 - 1/p):.6f}"
print(f"{p},{gx},{gy},{steps},{t:.6f},{mc:.1f},{sp:.4f},{eff:.4f},{kf}")
PY
  tail -1 "$CSV"
done
echo "wrote $CSV"
