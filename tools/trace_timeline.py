#!/usr/bin/env python3
"""tools/trace_timeline.py KERNEL_TRACE.csv [N] — timeline of the last N kernel dispatches of a
rocprofv3 --kernel-trace run: start (us, relative), duration, gap since the previous kernel ended,
queue, kernel name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    name = r["Kernel_Name"]
    name = name.split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end or 0, e)
