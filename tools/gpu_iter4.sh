#!/bin/bash
set -x
mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -12 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/sweep_variants.py --shape 16384x16384 4096x8192 8192x8192 --steps 25 --rounds 2 --variants 1 --ry 32 64 --pf 2 --fuse 4 5 6 --multistep 0 > gpurun_out/sweep_overlap.log 2>&1; echo "sweep rc=$?"
grep "^{" gpurun_out/sweep_overlap.log | python3 -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print(r['n'], r['ny'], 'ry', r['rows_per_chunk'], 'fuse', r['fuse'], 'ms', r['multistep'], 'ms/step %.4f' % r['ms_med'], 'Mcell/s %.0f' % r['mcells'])
"
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_iter.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_iter.log
