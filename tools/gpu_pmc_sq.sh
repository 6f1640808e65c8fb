#!/bin/bash
# where do the wave cycles of the 4-step kernel go: small tile vs full grid (SQ counters)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for shape in 4096x8192 16384x16384; do
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$shape -- python3 $R/tools/sweep_variants.py --shape $shape --steps 13 --rounds 1 --variants 1 --ry 64 --pf 2 --fuse 4 > $R/gpurun_out/pmc_sq_$shape.log 2>&1 || exit 1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for shape in ("4096x8192", "16384x16384"):
    for f in glob.glob(f"gpurun_out/pmc_sq_{shape}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            if "k_sweepT_dpp" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            print(shape, k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
