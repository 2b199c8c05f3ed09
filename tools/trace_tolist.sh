#!/bin/bash
# kernel table of ws_transform_to_list_device at N x N (rocprofv3 --kernel-trace --stats around tools/exp_tolist_device.py)
#   tools/trace_tolist.sh <N> <tag>  ->  gpurun_out/<tag>/tolist_<N>_kernel_stats.csv
n=${1:-8192}; tag=${2:-tolist}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/tools/exp_tolist_device.py $n > $out/tolist_$n.log 2>&1 || exit 1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/tolist_${n}_kernel_stats.csv; rm -rf $out/kt
grep "ws_transform_to_list_device" $out/tolist_$n.log
python3 - $out/tolist_${n}_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wsk::' in r['Name']:
        print('   %-78s calls %5s avg %9.1f us total %9.2f ms' % (r['Name'][:78], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
