#!/bin/bash
# HBM traffic and VALU counters of the DIFFUSION-ONLY flavour (vx = vy = 0) of k_sweepO_dpp at 16384^2, depths 6 and 7:
# FETCH_SIZE and WRITE_SIZE in separate process runs, SQ counters in a third (tools/gpu_pmc.sh does the same for the
# flavour bench.py times).  Output: gpurun_out/profiles_still/{pmc_traffic.json, sq_valu.json}
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
W="--physics 1.0,0.1,0,0 --bcs pppp --depths 6 7 --rows 110 182 230"
mkdir -p $R/gpurun_out/still
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/still/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/still/pmc_$c -- python3 $R/tools/pmc_workload.py $W > $R/gpurun_out/still/pmc_$c.log 2>&1 || exit 1
done
rm -rf $R/gpurun_out/still/pmc_SQ
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/still/pmc_SQ -- python3 $R/tools/pmc_workload.py $W > $R/gpurun_out/still/pmc_SQ.log 2>&1 || exit 1
cd $R
python3 tools/pmc_collect.py gpurun_out/still gpurun_out/profiles_still r03-diffusion-only || exit 1
# BASELINE configs[1] itself: 4096^2 (two 134 MB fields beside a 256 MiB Infinity Cache), the chunk height bench.py's trial picks there
W2="--nx 4096 --ny 4096 --physics 1.0,0.1,0,0 --bcs pppp --depths 7 --rows 38 --passes 12"
mkdir -p $R/gpurun_out/still4k
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/still4k/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/still4k/pmc_$c -- python3 $R/tools/pmc_workload.py $W2 > $R/gpurun_out/still4k/pmc_$c.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_collect.py gpurun_out/still4k gpurun_out/profiles_still4k r03-diffusion-only-4096 || exit 1
exit 0
