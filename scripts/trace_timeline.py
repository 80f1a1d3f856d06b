#!/usr/bin/env python3
"""Print the kernels of a rocprofv3 --kernel-trace csv as a timeline (start offset and duration in µs) — shows whether launches on two queues ran
side by side.  usage: trace_timeline.py <dir-or-csv> [name-substring] [last-n]"""
import csv, glob, os, sys
src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", ""))))
rows.sort()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [r for r in rows if sub in r[2]]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = rows[-last:]
t0 = rows[0][0]
for s, e, n, q in rows:
    print("%10.1f %10.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:70]))
print("span %.1f us, sum of durations %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3, sum(r[1] - r[0] for r in rows) / 1e3))
