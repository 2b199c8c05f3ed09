#!/usr/bin/env python3
"""Per-block averages of PMC counters over the launches of the late-pass relaxation kernel in the LAST transform of a
`rocprofv3 --pmc ... -- python3 tools/exp_smooth_trace.py` run.  usage: pmc_smooth_summary.py <counter_collection.csv>... [--block 20]"""
import csv, collections, sys
files = [a for a in sys.argv[1:] if not a.startswith("--")]
block = int(sys.argv[sys.argv.index("--block") + 1]) if "--block" in sys.argv else 20
for fn in files:
    rows = [r for r in csv.DictReader(open(fn)) if "k_relax<8, true, true, true, true>" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        d = by[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(by)
    # the last transform: the launches after the last gap in dispatch ids larger than 8 (other kernels between transforms)
    cut = 0
    for i in range(1, len(ids)):
        if ids[i] - ids[i - 1] > 8:
            cut = i
    ids = ids[cut:]
    print(f"# {fn}: {len(ids)} launches")
    for i in range(0, len(ids), block):
        blk = [by[k] for k in ids[i:i + block]]
        keys = [k for k in blk[0] if k != "dur"]
        print("launch %3d..  duration %6.1f us  " % (i, sum(b["dur"] for b in blk) / len(blk)) + "  ".join("%s %.4g" % (k, sum(b[k] for b in blk) / len(blk)) for k in keys))
