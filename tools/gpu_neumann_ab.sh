#!/bin/bash
# A/B of two builds of the engine on tiles with physical NEUMANN sides (tools/torus_bench.py --links/--bc):
#   new = climate-sim-mpi-cpp_amd/lib/libcsim.so, old = $OLD_LIB; interleaved twice so that box drift shows.
# Output: gpurun_out/neumann_ab.jsonl
R=${GRAFT_REPO_ROOT:-$PWD}
OLD=${OLD_LIB:?set OLD_LIB to the other build of libcsim.so}
out=$R/gpurun_out/neumann_ab.jsonl
: > $out
for rnd in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export CSIM_LIB=$OLD; else unset CSIM_LIB; fi
    for links in 1101 0101 1100; do
      for run in 20 0; do
        timeout -k 10 300 python3 $R/tools/torus_bench.py --shape 4096x8192 --steps 1200 --run $run --links $links --bc nnnn --modes torus-auto 2>/dev/null \
          | sed "s/^{/{\"lib\": \"$v\", \"round\": $rnd, /" >> $out || exit 1
      done
    done
  done
done
cat $out
