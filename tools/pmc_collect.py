#!/usr/bin/env python3
"""tools/pmc_collect.py — turns the rocprofv3 counter CSVs of tools/pmc_workload.py runs into
profiles/pmc_traffic.json (HBM bytes per launch per kernel configuration) and profiles/sq_valu.json
(VALU instructions per launch, sustained clock).  Usage: pmc_collect.py <gpurun_out dir> <profiles dir> <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sys


def sequence(log):
    for ln in open(log):
        if ln.startswith("PMC_SEQUENCE "):
            return json.loads(ln[len("PMC_SEQUENCE "):])
    raise SystemExit(f"no PMC_SEQUENCE line in {log}")


def sweep_rows(d):
    """[(dispatch id, kernel name, {counter: value})] of the sweep kernels, in dispatch order"""
    per = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_sweep" not in row["Kernel_Name"]:
                continue
            k = int(row["Dispatch_Id"])
            per.setdefault(k, [row["Kernel_Name"].split("(")[0].replace("void ", ""), {}])
            per[k][1][row["Counter_Name"]] = per[k][1].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(per.items())]


def durations(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return out


def attribute(rows, seq):
    """k-th sweep dispatch -> its entry of the printed launch sequence"""
    want = sum(s["launches"] for s in seq["sequence"])
    if len(rows) != want:
        raise SystemExit(f"profile has {len(rows)} sweep dispatches, the workload printed {want}")
    out, k = [], 0
    for s in seq["sequence"]:
        chunk = rows[k:k + s["launches"]]
        k += s["launches"]
        T = s["steps_per_launch"]
        for _, name, _ in chunk:
            m = re.search(r"k_sweepO_dpp<\d+, (\d+),", name)
            assert (T == 1 and "k_sweep_dpp" in name) or (m and int(m.group(1)) == T), (name, s)
        out.append((s, chunk))
    return out


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    acc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        seq = sequence(os.path.join(src, f"pmc_{c}.log"))
        for s, chunk in attribute(sweep_rows(os.path.join(src, f"pmc_{c}")), seq):
            key = (s["bc"], s["steps_per_launch"], s["rows_per_chunk"])
            vals = [cv[c] for _, _, cv in chunk]
            acc.setdefault(key, dict(kernel=chunk[0][1], nx=seq["nx"], ny=seq["ny"], bc=s["bc"],
                                     steps_per_launch=s["steps_per_launch"], rows_per_chunk=s["rows_per_chunk"]))
            acc[key][c + "_KiB_mean"] = sum(vals) / len(vals)
            acc[key][c + "_launches"] = len(vals)
    entries = []
    for key in sorted(acc):
        e = acc[key]
        if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024
            e["algorithmic_bytes_per_launch"] = e["nx"] * e["ny"] * 16
            e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
            entries.append(e)
    json.dump(dict(tag=tag, collected_by="tools/gpu_pmc.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per "
                   "process run of tools/pmc_workload.py (the two do not fit one pass)",
                   fetch_correction="x2: on gfx950 FETCH_SIZE reports 1/2 of a 16-B-per-lane streaming read "
                                    "(MI355X_MICROARCH.md, HBM)",
                   unit_note="FETCH_SIZE / WRITE_SIZE count KiB; hbm_bytes_per_launch = (2 x FETCH + WRITE) x 1024",
                   entries=entries), open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print(f"pmc_traffic.json: {len(entries)} entries")
    for e in entries:
        print(f"  {e['bc']} T={e['steps_per_launch']} ry={e['rows_per_chunk']:4d}  {e['hbm_bytes_per_launch'] / 1e9:.3f} GB "
              f"= {e['traffic_over_algorithmic']:.3f} x algorithmic")
    sq_log = os.path.join(src, "pmc_SQ.log")
    if os.path.exists(sq_log):
        seq = sequence(sq_log)
        dur = durations(os.path.join(src, "pmc_SQ"))
        out = []
        for s, chunk in attribute(sweep_rows(os.path.join(src, "pmc_SQ")), seq):
            chunk = chunk[1:] or chunk   # the first launch of a configuration runs on cold clocks / caches
            m = collections.defaultdict(float)
            for did, _, cv in chunk:
                for n, v in cv.items():
                    m[n] += v / len(chunk)
                m["ns"] += dur.get(did, 0) / len(chunk)
            T = s["steps_per_launch"]
            e = dict(kernel=chunk[0][1], nx=seq["nx"], ny=seq["ny"], bc=s["bc"], steps_per_launch=T,
                     rows_per_chunk=s["rows_per_chunk"], kernel_us_under_counters=m["ns"] / 1e3,
                     clock_ghz=(m["GRBM_GUI_ACTIVE"] / 8 / m["ns"]) if m["ns"] else None,
                     fp64_share=(14.0 * 2 * T) / (14.0 * 2 * T + 4 * T + 2) if T > 1 else 30.0 / 36.0)
            for n in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
                      "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
                if n in m:
                    e[n] = m[n]
            out.append(e)
        # bench.py looks an entry up by T: keep, per T, the all-Dirichlet one with the middle chunk height
        best = {}
        for e in out:
            if e["bc"] == "dddd":
                best.setdefault(e["steps_per_launch"], []).append(e)
        entries = [v[len(v) // 2] for _, v in sorted(best.items())]
        json.dump(dict(tag=tag, collected_by="tools/gpu_pmc.sh: rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE --kernel-trace on "
                       "tools/pmc_workload.py", note="SQ_INSTS_VALU = wave-level VALU instructions per launch; "
                       "clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same (counter-slowed) run; "
                       "fp64_share = fp64 add/mul/fma among the VALU instructions of the steady-state body "
                       "(28 T fp64 — E - 2c and N - 2c are one fma each — + 4 T DPP moves + 2 screening compares "
                       "per lane-row at depth T)",
                       entries=entries, all=out), open(os.path.join(dst, "sq_valu.json"), "w"), indent=1)
        for e in entries:
            print(f"  SQ T={e['steps_per_launch']} ry={e['rows_per_chunk']} VALU insts {e.get('SQ_INSTS_VALU', 0) / 1e6:.1f} M "
                  f"clock {e['clock_ghz']:.2f} GHz  {e['kernel_us_under_counters']:.0f} us")


if __name__ == "__main__":
    main()
