#!/bin/bash
# rocprofv3 evidence for the bench command: kernel-trace stats, then HBM traffic of the sweep
# kernels from PMC counters, one counter per pass (guide: FETCH_SIZE and WRITE_SIZE do not fit
# one pass; FETCH_SIZE reads 1/2 of wide streaming reads on gfx950)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
timeout -k 10 600 python3 $R/bench.py --steps 100 --warmup 10 > $R/gpurun_out/bench_full.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 14 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
cd $R
tail -1 gpurun_out/bench_full.log
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv"); do head -8 $f; done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                acc[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(c, k, "n=", len(v), "mean=", sum(v)/len(v))
PY
