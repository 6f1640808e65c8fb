#!/bin/bash
# tools/bench_lines.sh — the bench lines kept under profiles/ (run on the GPU box through gpurun; copy gpurun_out/lines/*.json to profiles/)
set -e
mkdir -p gpurun_out/lines
python bench.py > gpurun_out/lines/r03_bench_line.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/lines/r03_bench_line_steps20.json 2>/dev/null
python bench.py --bc nnnn --no-cpu-baseline > gpurun_out/lines/r03_bench_line_nnnn.json 2>/dev/null
python bench.py --bc dnpd --no-cpu-baseline > gpurun_out/lines/r03_bench_line_dnpd.json 2>/dev/null
python bench.py --contract 1 --no-cpu-baseline > gpurun_out/lines/r03_bench_line_contract.json 2>/dev/null
CSIM_BENCH_SELF_TORUS=1 python bench.py --nx 4096 --ny 8192 --no-cpu-baseline > gpurun_out/lines/r03_bench_selftorus_4096x8192.json 2>/dev/null
CSIM_BENCH_SELF_TORUS=1 python bench.py --nx 4096 --ny 8192 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/lines/r03_bench_selftorus_4096x8192_steps20.json 2>/dev/null
for f in gpurun_out/lines/*.json; do python3 -c "
import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['roofline']['frac'] if d.get('roofline') else None, d['roofline'].get('kernel_avg_ms') if d.get('roofline') else None)"; done
# kernel stats of the default command (rocprofv3 summary that profiles/r03_bench_kernel_stats.csv holds)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
rm -rf $R/gpurun_out/prof_stats2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats2 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats2.log 2>&1
cd $R
for f in $(find gpurun_out/prof_stats2 -name "*kernel_stats.csv"); do cp $f gpurun_out/lines/r03_bench_kernel_stats.csv; head -4 $f; done
grep "^{" gpurun_out/prof_stats2.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprof:', round(d['value']), d['roofline']['kernel_avg_ms'], d['roofline']['launches_timed'])"
