#!/bin/bash
# tools/rccl_env_ab.sh — does an RCCL setting change what the exchange costs beside the sweep?  The merged-launch
# torus pass (tools/torus_bench.py) under a few NCCL_* environments, one process each, default first and last.
mkdir -p gpurun_out
OUT=gpurun_out/rccl_env_ab.jsonl
: > $OUT
run() {  # label, env assignments...
    local label=$1; shift
    env "$@" timeout -k 10 120 python3 tools/torus_bench.py --shape ${SHAPE:-4096x8192} --steps ${STEPS:-12000} --modes torus-merged 2>&1 \
        | grep '"tile"' | sed "s/^{/{\"env\": \"$label\", /" >> $OUT || return 1
}
run default X=1 &&
run nchannels_per_peer=1 NCCL_NCHANNELS_PER_PEER=1 &&
run nchannels_per_peer=4 NCCL_NCHANNELS_PER_PEER=4 &&
run nchannels_per_peer=8 NCCL_NCHANNELS_PER_PEER=8 &&
run proto=LL NCCL_PROTO=LL &&
run proto=LL128 NCCL_PROTO=LL128 &&
run proto=Simple NCCL_PROTO=Simple &&
run max_nchannels=4 NCCL_MAX_NCHANNELS=4 &&
run min_nchannels=32 NCCL_MIN_NCHANNELS=32 &&
run default X=1
cat $OUT
