#!/bin/bash
# rocprofv3 evidence for the bench command: the bench line, kernel-trace stats of the same command,
# then HBM traffic of the sweep kernels from PMC counters, one counter per pass (guide: FETCH_SIZE
# and WRITE_SIZE do not fit one pass; FETCH_SIZE reads 1/2 of wide streaming reads on gfx950).
# Writes gpurun_out/{bench_full.log, prof_stats/, pmc_*/, pmc_traffic.json}.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
RY=${RY:-122}
cd /tmp
timeout -k 10 900 python3 $R/bench.py > $R/gpurun_out/bench_full.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 26 --warmup 6 --ramp-seconds 0 --rows-per-chunk $RY --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
cd $R
tail -1 gpurun_out/bench_full.log
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv"); do head -8 $f; done
RY=$RY python3 - <<'PY'
import csv, glob, collections, json, os
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        a = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                a[row["Kernel_Name"].split("(")[0][-40:]].append(float(row["Counter_Value"]))
        for k, v in a.items():
            print(c, k, "n=", len(v), "mean=", sum(v) / len(v))
            acc.setdefault(k, {})[c] = sum(v) / len(v)
out = {"nx": 16384, "ny": 16384, "rows_per_chunk": int(os.environ["RY"]),
       "fetch_correction": "x2: on gfx950 FETCH_SIZE reports 1/2 of a 16-B-per-lane streaming read (MI355X_MICROARCH.md, HBM)",
       "unit_note": "FETCH_SIZE / WRITE_SIZE count KiB", "kernels": {}}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = dict(FETCH_SIZE_KiB_mean=v["FETCH_SIZE"], WRITE_SIZE_KiB_mean=v["WRITE_SIZE"],
                                 hbm_bytes_per_launch=(2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
for k, v in out["kernels"].items():
    if "k_sweepO_dpp<0, 6" in k:
        out.update(kernel=k, steps_per_launch=6, hbm_bytes_per_launch=v["hbm_bytes_per_launch"],
                   algorithmic_bytes_per_launch=16384 * 16384 * 16 * 6)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
