#!/bin/bash
# first GPU contact: runtime sanity in both loader modes, parity tests, short bench, variant sweep
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
CSIM_PRELOAD_TORCH=0 timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_notorch.log 2>&1; echo "smoke_notorch rc=$?"
timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_torch.log 2>&1; echo "smoke_torch rc=$?"
tail -3 gpurun_out/smoke_notorch.log gpurun_out/smoke_torch.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_first.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/bench_first.log
timeout -k 10 600 python tools/sweep_variants.py --n 16384 --steps 10 --rounds 2 --out gpurun_out/sweep_16384.json > gpurun_out/sweep_16384.log 2>&1; echo "sweep rc=$?"
tail -40 gpurun_out/sweep_16384.log
