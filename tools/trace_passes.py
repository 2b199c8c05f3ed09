#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints, for the LAST transform in it, every launch in order
with its duration: which relaxation pass costs what.  usage: trace_passes.py <..._kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a transform starts at the seed scatter
starts = [i for i, n in enumerate(names) if "k_paint_sorted" in n or "k_seed_tables" in n]
if not starts:
    sys.exit("no seed kernel launch in trace")
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = r["Kernel_Name"].split("(")[0][:48]
    print(f"+{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>9}  {short}")
