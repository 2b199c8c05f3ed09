#!/usr/bin/env python3
"""Launch by launch, the LAST transform of a rocprofv3 --kernel-trace CSV (from its last k_seed_tables on): kernel, duration.
usage: trace_last_transform.py <..._kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wsk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_seed_tables" in r["Kernel_Name"]][-1]
prev = None
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev is None else (s - prev) / 1e3
    print(f"{r['Kernel_Name'].split('(')[0][:64]:64s} {(e - s) / 1e3:8.1f} us  gap {gap:5.1f}")
    prev = e
