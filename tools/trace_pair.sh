#!/bin/bash
# kernel table of the call pair as one call / as two: tools/trace_pair.sh <tag>  ->  gpurun_out/<tag>/pair_{one,two}_kernel_stats.csv
tag=${1:-pair}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for form in one two; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/tools/exp_pair.py 8192 10 $form > $out/pair_$form.log 2>&1 || exit 1
  f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/pair_${form}_kernel_stats.csv; rm -rf $out/kt
  grep "call(s)" $out/pair_$form.log
  python3 - $out/pair_${form}_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wsk::' in r['Name']:
        print('   %-70s calls %4s avg %8.1f us' % (r['Name'].replace('void ','').replace('wsk::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
