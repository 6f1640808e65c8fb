#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/pytest_comm.log 2>&1; echo "comm rc=$?"
tail -15 gpurun_out/pytest_comm.log
timeout -k 10 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_r01.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_r01.log
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r01 -name "*stats*" | head; for f in $(find gpurun_out/prof_r01 -name "*kernel_stats.csv"); do head -12 $f; done
