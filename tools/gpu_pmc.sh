#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box through gpurun):
#   1. the bench line, default flags                                -> gpurun_out/bench_full.log
#   2. kernel-trace stats of the same command                        -> gpurun_out/prof_stats/
#   3. HBM traffic of every sweep instantiation bench.py can time, from PMC counters: FETCH_SIZE and
#      WRITE_SIZE in SEPARATE process runs (they do not fit one pass), SQ counters in a third
#   4. tools/pmc_collect.py -> gpurun_out/profiles_new/{pmc_traffic.json, sq_valu.json}
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r03}
cd /tmp
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout -k 10 900 python3 $R/bench.py > $R/gpurun_out/bench_full.log 2> $R/gpurun_out/bench_full.err || exit 1
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1 || exit 1
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/pmc_workload.py > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
rm -rf $R/gpurun_out/pmc_SQ
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_SQ -- python3 $R/tools/pmc_workload.py --bcs dddd > $R/gpurun_out/pmc_SQ.log 2>&1 || exit 1
cd $R
python3 tools/pmc_collect.py gpurun_out gpurun_out/profiles_new $TAG || exit 1
[ -f gpurun_out/bench_full.log ] && tail -1 gpurun_out/bench_full.log
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv" 2>/dev/null); do head -6 $f; done
exit 0
