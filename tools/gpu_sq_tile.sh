#!/bin/bash
# SQ counters (VALU instructions, busy cycles, sustained clock) of the multi-step sweep on a per-GPU tile, next to the
# bench grid: is a single round of wavefronts slower per instruction, or does it execute more of them?
# Output: gpurun_out/sq_tile/{tile,full}.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/sq_tile
cd /tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for spec in "tile --nx 4096 --ny 8192 --bcs dddd --depths 6 --rows 74 --passes 8" "full --nx 16384 --ny 16384 --bcs dddd --depths 6 --rows 182 --passes 4"; do
  name=${spec%% *}; args=${spec#* }
  rm -rf $R/gpurun_out/sq_tile/$name
  mkdir -p $R/gpurun_out/sq_tile/$name
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq_tile/$name/pmc_SQ -- python3 $R/tools/pmc_workload.py $args > $R/gpurun_out/sq_tile/$name/pmc_SQ.log 2>&1 || exit 1
done
cd $R
python3 - <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
import pmc_collect as pc
for name in ("tile", "full"):
    d = os.path.join("gpurun_out", "sq_tile", name)
    seq = pc.sequence(os.path.join(d, "pmc_SQ.log"))
    dur = pc.durations(os.path.join(d, "pmc_SQ"))
    for s, chunk in pc.attribute(pc.sweep_rows(os.path.join(d, "pmc_SQ")), seq):
        chunk = chunk[1:] or chunk
        n = len(chunk)
        m = {}
        for did, _, cv in chunk:
            for k, v in cv.items():
                m[k] = m.get(k, 0.0) + v / n
            m["ns"] = m.get("ns", 0.0) + dur.get(did, 0) / n
        cells = seq["nx"] * seq["ny"] * s["steps_per_launch"]
        print(name, seq["nx"], seq["ny"], "T", s["steps_per_launch"], "rows", s["rows_per_chunk"], "us", round(m["ns"] / 1e3, 1),
              "VALU insts/cell-step", round(m["SQ_INSTS_VALU"] * 64 / cells, 2), "clock GHz", round(m["GRBM_GUI_ACTIVE"] / 8 / m["ns"], 3),
              "waves", int(m["SQ_WAVES"]), "VALU active/busy", round(m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_BUSY_CYCLES"], 3) if m.get("SQ_BUSY_CYCLES") else None,
              "wait_inst_any/wave_cycles", round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3))
PY
