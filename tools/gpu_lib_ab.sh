#!/bin/bash
# A/B of two builds of the engine on the per-GPU tiles (tools/torus_bench.py) and on the bench grid:
#   new = climate-sim-mpi-cpp_amd/lib/libcsim.so, old = $OLD_LIB (another build of the engine)
# interleaved twice so that box drift shows.  Output: gpurun_out/lib_ab.jsonl
R=${GRAFT_REPO_ROOT:-$PWD}
OLD=${OLD_LIB:?set OLD_LIB to the other build of libcsim.so (e.g. make OUT=../lib_old/libcsim.so OBJDIR=../build_old in a checkout of the other revision)}
out=$R/gpurun_out/lib_ab.jsonl
: > $out
for rnd in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export CSIM_LIB=$OLD; else unset CSIM_LIB; fi
    for run in 20 0; do
      timeout -k 10 300 python3 $R/tools/torus_bench.py --shape 4096x8192 8192x8192 --steps 1200 --run $run --modes single torus-auto 2>/dev/null \
        | sed "s/^{/{\"lib\": \"$v\", \"round\": $rnd, /" >> $out || exit 1
    done
    timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$v', 'round': $rnd, 'bench': '16384x16384 --steps 20', 'mcells': d['value'], 'repeats_ms_per_step': d['config'].get('repeats_ms_per_step')}))" >> $out || exit 1
  done
done
cat $out
