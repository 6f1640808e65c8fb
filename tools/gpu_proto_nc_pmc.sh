#!/bin/bash
# Why the 4-columns-per-lane form of the overlapped-strip sweep was not productised: counters of the prototype
# (tools/proto/proto_nc.hip: same march, T = 6, 2 vs 4 columns per lane, interior body only) on 16384^2.
#   pass 1: rocprofv3 --kernel-trace --stats          -> durations without counters
#   pass 2: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# Output: gpurun_out/proto_nc_pmc.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
[ -x $R/tools/proto/proto_nc ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $R/tools/proto/proto_nc $R/tools/proto/proto_nc.hip || exit 1
rm -rf $R/gpurun_out/pnc_trace $R/gpurun_out/pnc_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pnc_trace -- $R/tools/proto/proto_nc > $R/gpurun_out/pnc_trace.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pnc_pmc -- $R/tools/proto/proto_nc > $R/gpurun_out/pnc_pmc.log 2>&1 || exit 1
cd $R
python3 - <<'PY' > gpurun_out/proto_nc_pmc.txt
import collections, csv, glob
def trace(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]].append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out
t0 = trace("gpurun_out/pnc_trace")
t1 = trace("gpurun_out/pnc_pmc")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pnc_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# tools/proto/proto_nc.hip on 16384^2, T = 6, interior body only: 2 vs 4 columns per lane (means over all launches, all chunk heights)")
print(open("gpurun_out/pnc_trace.log").read().strip().split("W2026")[0].strip())
for k in sorted(t0):
    d0 = [x[1] for x in t0[k]]
    d1 = [x[1] for x in t1.get(k, [])]
    c = {n: sum(v) / len(v) for n, v in cnt.get(k, {}).items()}
    us0, us1 = sum(d0) / len(d0) / 1e3, (sum(d1) / len(d1) / 1e3 if d1 else 0)
    line = f"{k}: {len(d0)} launches, {us0:.1f} us without counters, {us1:.1f} us under counters"
    if c:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (us1 * 1e3) if us1 else 0
        busy = c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 / 4 * 1024)
        line += (f"; SQ_INSTS_VALU {c['SQ_INSTS_VALU'] / 1e6:.1f} M, SQ_WAVES {c['SQ_WAVES']:.0f}, VALU busy {busy:.3f} of the SIMD-quad-cycles, "
                 f"clock {clk:.2f} GHz, wave-cycles per wave {c['SQ_WAVE_CYCLES'] / c['SQ_WAVES']:.0f}")
    print(line)
PY
cat gpurun_out/proto_nc_pmc.txt
