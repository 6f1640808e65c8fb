#!/bin/bash
# kernel timelines of 20-step calls (7 + 7 + 6) on the 8-GPU tile: self-linked torus under the default schedule
# (bulk-first on short runs), the merged schedule, and the tile without neighbours
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for m in torus-auto torus-merged single; do
  MODE=$m RUN=20 N=60 SHAPE=${SHAPE:-4096x8192} bash tools/gpu_trace_torus.sh || exit 1
  echo "== $m"; tail -40 gpurun_out/timeline_$m.txt | cut -c1-150
done
