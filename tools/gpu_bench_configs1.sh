#!/bin/bash
# bench.py on BASELINE configs[1] (4096^2, diffusion only, all Periodic) and the same physics on the bench grid, after
# refreshing the checksum fixture the parity preflight compares with (tests/golden/bench_checksum.json: copy back from
# gpurun_out/ if it changed).  Output: gpurun_out/r3_bench_c2.json, gpurun_out/r3_bench_still16k.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python3 tools/make_bench_checksum.py > gpurun_out/r3_make_checksum.log 2>&1 || { tail -5 gpurun_out/r3_make_checksum.log; exit 1; }
cp gpurun_out/bench_checksum.json tests/golden/bench_checksum.json
python3 bench.py --nx 4096 --ny 4096 --bc pppp --physics 1.0,0.1,0,0 > gpurun_out/r3_bench_c2.json 2> gpurun_out/r3_bench_c2.err || { tail -5 gpurun_out/r3_bench_c2.err; exit 1; }
python3 bench.py --bc pppp --physics 1.0,0.1,0,0 --no-cpu-baseline > gpurun_out/r3_bench_still16k.json 2> gpurun_out/r3_bench_still16k.err || { tail -5 gpurun_out/r3_bench_still16k.err; exit 1; }
for f in gpurun_out/r3_bench_c2.json gpurun_out/r3_bench_still16k.json; do tail -1 $f | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(round(d['value']), d['config'].get('repeats_ms_per_step'), 'frac', round(r['frac'], 3), 'binding', r['is_binding'], 'traffic', r['traffic'], r['kernel'][:70], 'kern ms', round(r['kernel_avg_ms'], 4), d['config']['parity_preflight']['schedules'], (d.get('cpu_baseline') or {}).get('value'))
"; done
