export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
rm -rf $R/gpurun_out/trace20
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/trace20.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/trace20 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sw=[r for r in rows if 'k_sweepO' in r['Kernel_Name']]
t_prev_end=None
for r in sw[-12:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('<')[1].split('>')[0]
    print(name, 'dur_us', round((e-s)/1e3,1), 'gap_us', None if t_prev_end is None else round((s-t_prev_end)/1e3,1))
    t_prev_end=e
PY
grep "^{" gpurun_out/trace20.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), d[\"ms_per_step\"]*20, d[\"roofline\"][\"kernel_avg_ms\"])"
