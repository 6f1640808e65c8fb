R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r3_relay_tests.log 2>&1 || { tail -30 gpurun_out/r3_relay_tests.log; exit 1; }
tail -2 gpurun_out/r3_relay_tests.log
: > gpurun_out/relay_ab.jsonl
for rnd in 1 2; do
  for run in 20 0; do
    timeout -k 10 300 python3 tools/torus_bench.py --shape 4096x8192 8192x8192 --steps 1200 --run $run --modes single torus-auto torus-auto+relay=0 torus-bulkfirst torus-bulkfirst+relay=0 torus-merged 2>/dev/null | grep "^{" >> gpurun_out/relay_ab.jsonl || exit 1
  done
done
cat gpurun_out/relay_ab.jsonl
