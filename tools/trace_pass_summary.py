#!/usr/bin/env python3
"""Condensed view of the LAST transform in a rocprofv3 kernel-trace CSV: per launch start offset, duration and the gap
to the previous launch, aggregated in blocks of 16 launches (a smooth map runs hundreds of passes).
usage: trace_pass_summary.py <..._kernel_trace.csv> [block=16]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
block = int(sys.argv[2]) if len(sys.argv) > 2 else 16
starts = [i for i, r in enumerate(rows) if "k_seed_tables" in r["Kernel_Name"] or "k_paint_sorted" in r["Kernel_Name"]]
lo = starts[-1]
sel = rows[lo:]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = None
acc = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    acc.append((s - t0, e - s, gap, r["Kernel_Name"].split("(")[0][:40]))
    prev_end = e
tot_k = sum(a[1] for a in acc); tot_g = sum(max(a[2], 0) for a in acc)
print(f"launches {len(acc)}  kernel time {tot_k/1e3:.1f} us  gaps {tot_g/1e3:.1f} us  span {(acc[-1][0]+acc[-1][1])/1e3:.1f} us")
for i in range(0, len(acc), block):
    b = acc[i:i + block]
    print(f"launch {i:4d}..{i+len(b)-1:4d}  at +{b[0][0]/1e3:9.1f} us  kernels {sum(x[1] for x in b)/1e3:8.1f} us (max {max(x[1] for x in b)/1e3:6.1f})  "
          f"gaps {sum(max(x[2],0) for x in b)/1e3:7.1f} us  first: {b[0][3]}")
