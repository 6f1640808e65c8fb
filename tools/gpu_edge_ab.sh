#!/bin/bash
# parity of the new edge body + counters old/new
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r3_edge_parity.log 2>&1 || { tail -30 gpurun_out/r3_edge_parity.log; exit 1; }
tail -3 gpurun_out/r3_edge_parity.log
cd /tmp
# "old" = another build of the engine (OLD_LIB=/path/to/libcsim.so, e.g. `make OUT=../lib_old/libcsim.so OBJDIR=../build_old` in a
# checkout of the other revision); without it only the current build is measured
for v in new ${OLD_LIB:+old}; do
  if [ $v = old ]; then export CSIM_LIB=$OLD_LIB; else unset CSIM_LIB; fi
  rm -rf $R/gpurun_out/edge_$v
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/edge_$v -- python3 $R/tools/edge_workload.py > $R/gpurun_out/edge_$v.log 2>&1 || { tail -20 $R/gpurun_out/edge_$v.log; exit 1; }
  python3 $R/tools/edge_collect.py $R/gpurun_out/edge_$v $R/gpurun_out/edge_$v.log > $R/gpurun_out/edge_$v.txt
  cat $R/gpurun_out/edge_$v.txt
done
