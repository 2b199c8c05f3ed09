#!/bin/bash
# per-level durations of the two per-level kernels of ws_transform_to_list_device at N x N (last transform of the run)
#   tools/trace_tolist_levels.sh <N> <tag>  ->  gpurun_out/<tag>/tolist_<N>_levels.txt
n=${1:-8192}; tag=${2:-tolist}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/ktl -- python3 $root/tools/exp_tolist_device.py $n > $out/tolist_levels_$n.log 2>&1 || exit 1
f=$(find $out/ktl -name "*kernel_trace.csv" | head -1)
python3 - $f > $out/tolist_${n}_levels.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wsk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_seed_tables" in r["Kernel_Name"]][-1]
last = rows[idx:]
t0 = int(last[0]["Start_Timestamp"])
per = {}
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("wsk::", "")
    per.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("span of the last transform: %.2f ms" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e6))
for name, d in per.items():
    print("%-40s launches %4d total %9.1f us" % (name[:40], len(d), sum(d)))
for name, d in per.items():
    if len(d) >= 200:
        print(name, "by launch (us), 16 per row:")
        for i in range(0, len(d), 16):
            print("  %3d: " % i + " ".join("%6.1f" % x for x in d[i:i + 16]))
PY
rm -rf $out/ktl
cat $out/tolist_${n}_levels.txt | head -60
