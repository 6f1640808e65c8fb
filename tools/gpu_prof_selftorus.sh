#!/bin/bash
# rocprofv3 kernel stats of bench.py's N > 1 path on one GPU (self-linked torus, 8-GPU tile 4096 x 8192)
mkdir -p gpurun_out
export TMPDIR=/tmp
export CSIM_BENCH_SELF_TORUS=1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
rm -rf $R/gpurun_out/prof_selftorus
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_selftorus -- python3 $R/bench.py --nx ${NX:-4096} --ny ${NY:-8192} > $R/gpurun_out/prof_selftorus.log 2>&1 || exit 1
cd $R
tail -1 gpurun_out/prof_selftorus.log | cut -c1-300
for f in $(find gpurun_out/prof_selftorus -name "*kernel_stats.csv"); do head -9 $f | cut -c1-200; done
