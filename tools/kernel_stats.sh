#!/bin/bash
# kernel table of bench.py --steps 5 (headline, no extras): tools/kernel_stats.sh <tag>  ->  gpurun_out/<tag>_kernel_stats.csv + a short listing
tag=${1:-ks}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-extras --no-pipeline > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
f=$(find $out/kt_$tag -name "*kernel_stats.csv" | head -1); cp $f $out/${tag}_kernel_stats.csv; rm -rf $out/kt_$tag
python3 - $out/${tag}_kernel_stats.csv <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if 'wsk::' in r['Name'] and ('relax' in r['Name'] or 'resolve' in r['Name'] or 'seed_tables' in r['Name']):
        print('   %-64s calls %4s avg %8.1f us' % (r['Name'].replace('void ','').replace('wsk::','')[:64], r['Calls'], float(r['AverageNs'])/1e3))
PY
python3 -c "
import json,sys
d=json.loads(open('$out/${tag}_bench.json').read()); print('ms_per_step (one context, no pipeline):', d['ms_per_step'])"
