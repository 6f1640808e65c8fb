#!/bin/bash
set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -8 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/sweep_variants.py --n 16384 --steps 25 --rounds 2 --variants 1 --ry 64 128 256 --pf 2 --fuse 2 3 4 --out gpurun_out/sweepT2_16384.json > gpurun_out/sweepT2_16384.log 2>&1; echo "sweep rc=$?"
cat gpurun_out/sweepT2_16384.log | cut -c1-220
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_iter.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench_iter.log
