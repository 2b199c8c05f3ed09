#!/bin/bash
# kernel table of the 8192^2 merging transform (rocprofv3 --kernel-trace --stats around tools/exp_merge_time.py)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/mtrace; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/tools/exp_merge_time.py 8192 > $out/run.log 2>&1 || exit 1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; rm -rf $out/kt
grep "merge " $out/run.log
python3 - $out/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wsk::' in r['Name']:
        print('   %-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
