#!/bin/bash
# kernel timeline of the multi-rank pass (self-linked torus on one GPU): MODE=torus-overlap|torus-serial|single
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
MODE=${MODE:-torus-overlap}
cd /tmp
rm -rf $R/gpurun_out/trace_$MODE
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$MODE -- python3 $R/tools/torus_trace.py $MODE ${SHAPE:-4096x8192} > $R/gpurun_out/trace_$MODE.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/trace_$MODE -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f ${N:-40} > gpurun_out/timeline_$MODE.txt
tail -1 gpurun_out/trace_$MODE.log
