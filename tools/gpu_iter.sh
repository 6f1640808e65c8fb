#!/bin/bash
# quick GPU iteration: parity tests, then a short variant sweep and bench
set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -12 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/sweep_variants.py --n 16384 --steps 25 --rounds 2 --variants 1 --ry 32 64 128 256 --pf 2 4 --fuse 2 3 4 --out gpurun_out/sweepT_16384.json > gpurun_out/sweepT_16384.log 2>&1; echo "sweep rc=$?"
cat gpurun_out/sweepT_16384.log | cut -c1-220
timeout -k 10 300 python tools/sweep_variants.py --n 4096 8192 --steps 25 --rounds 2 --variants 1 --ry 64 128 --pf 2 --fuse 0 2 3 4 > gpurun_out/sweepT_small.log 2>&1; echo "sweep small rc=$?"
cat gpurun_out/sweepT_small.log | cut -c1-220
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_iter.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench_iter.log
