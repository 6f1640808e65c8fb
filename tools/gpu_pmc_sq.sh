#!/bin/bash
# where do the wave cycles of the dominant sweep kernel go (SQ counters) and what clock does the
# chip hold under it (GRBM_GUI_ACTIVE / 8 / kernel time)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for shape in ${SHAPES:-16384x16384}; do
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$shape -- python3 $R/tools/sweep_variants.py --shape $shape --steps 24 --rounds 1 --variants 1 --ry ${RY:-122} --pf 2 --fuse ${FUSE:-6} > $R/gpurun_out/pmc_sq_$shape.log 2>&1 || exit 1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in glob.glob("gpurun_out/pmc_sq_*x*"):
    if not os.path.isdir(d): continue
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            if "k_sweepO_dpp" in row["Kernel_Name"] or "k_sweepT_dpp" in row["Kernel_Name"]:
                k = row["Kernel_Name"][:34]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[k]["ns"].append(dur.get(row["Dispatch_Id"], (0,))[0])
        for k, c in acc.items():
            m = {n: sum(v) / len(v) for n, v in c.items()}
            ns = m.pop("ns")
            print(d, k, "n=", len(c["SQ_WAVES"]), "kernel_us=%.1f" % (ns / 1e3),
                  "clock_GHz=%.2f" % (m.get("GRBM_GUI_ACTIVE", 0) / 8 / ns) if ns else "", {n: round(v) for n, v in m.items()})
PY
