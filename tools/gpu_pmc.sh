#!/bin/bash
# HBM traffic of the sweep kernels from PMC counters, one counter per pass (guide: FETCH_SIZE
# and WRITE_SIZE do not fit one pass; FETCH_SIZE reads 1/2 of wide streaming reads on gfx950)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f | head -20
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(c, k, "n=", len(v), "mean=", sum(v)/len(v), "min=", min(v), "max=", max(v))
PY
