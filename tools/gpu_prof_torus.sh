#!/bin/bash
# per-kernel times of the multi-rank pass sequence (self-linked torus on one GPU)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_torus -- python3 $R/tools/torus_bench.py --shape ${SHAPE:-4096x8192} --steps 61 > $R/gpurun_out/prof_torus.log 2>&1 || exit 1
cd $R
for f in $(find gpurun_out/prof_torus -name "*kernel_stats.csv"); do cut -c1-200 $f | head -14; done
grep "^{" gpurun_out/prof_torus.log
