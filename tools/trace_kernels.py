#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints every launch after the LAST launch of the marker kernel
(argv[2], a substring) with start offset and duration.  usage: trace_kernels.py <csv> <marker> [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if not starts:
    sys.exit(f"no {marker} launch in trace")
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + limit]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"+{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:60]}")
