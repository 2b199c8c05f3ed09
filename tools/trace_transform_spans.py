#!/usr/bin/env python3
"""Every transform of a rocprofv3 --kernel-trace CSV (from one k_seed_tables to the next): launches, span, busy time, the gap
before the next transform -- and the launches of one of them (argv[2], counted from the end; default 12).
usage: trace_transform_spans.py <..._kernel_trace.csv> [k]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wsk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_seed_tables" in r["Kernel_Name"]] + [len(rows)]
for n, (a, b) in enumerate(zip(idx[:-1], idx[1:])):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    nxt = (int(rows[b]["Start_Timestamp"]) - t1) / 1e3 if b < len(rows) else 0.0
    print(f"transform {n:3d} ({n - len(idx) + 1:4d}): {len(seg):3d} launches, span {(t1 - t0) / 1e3:7.1f} us, busy {busy / 1e3:7.1f} us, then {nxt:7.1f} us")
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = idx[-1 - k], idx[-k]
prev = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {r['Kernel_Name'].split('(')[0][:60]:60s} {(e - s) / 1e3:8.1f} us  gap {0 if prev is None else (s - prev) / 1e3:5.1f}")
    prev = e
