#!/usr/bin/env python3
"""tools/edge_collect.py <rocprofv3 output dir> <log with EDGE_SEQUENCE> — executed VALU instructions and duration per
launch of tools/edge_workload.py's configurations, edge body relative to the interior body of the same depth."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, log = sys.argv[1], sys.argv[2]
    seq = None
    for ln in open(log):
        if ln.startswith("EDGE_SEQUENCE "):
            seq = json.loads(ln[len("EDGE_SEQUENCE "):])
    per = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_sweepO" not in row["Kernel_Name"]:
                continue
            k = int(row["Dispatch_Id"])
            per.setdefault(k, {})
            per[k][row["Counter_Name"]] = per[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    ids = sorted(per)
    want = sum(s["launches"] for s in seq["sequence"])
    assert len(ids) == want, (len(ids), want)
    k, base = 0, {}
    print(f"tile {seq['nx']} x {seq['ny']}: SQ_INSTS_VALU (wave-level) and duration per launch")
    for s in seq["sequence"]:
        chunk = ids[k:k + s["launches"]]
        k += s["launches"]
        insts = sum(per[i].get("SQ_INSTS_VALU", 0.0) for i in chunk) / len(chunk)
        us = sum(dur.get(i, 0) for i in chunk) / len(chunk) / 1e3
        if s["body"] == "interior":
            base[s["T"]] = (insts, us)
        bi, bu = base[s["T"]]
        print(f"  T={s['T']} {s['body']:9s} bc={s['bc']:5s} rows={s['rows']:4d}  VALU {insts / 1e6:8.2f} M = {insts / bi:5.3f} x interior   "
              f"{us:8.1f} us = {us / bu:5.3f} x interior")


if __name__ == "__main__":
    main()
