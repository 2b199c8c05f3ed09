#!/bin/bash
# kernel table of one 8192^2 field in 2 x 2 tiles on a local group of four (bench.py --config c5 --tiles 2x2): tools/trace_tiles2d.sh <tag>
tag=${1:-t2d}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --config c5 --size 8192 --local-ranks 4 --tiles 2x2 --no-extras --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err || exit 1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; rm -rf $out/kt
python3 - $out/kernel_stats.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'wsk::' in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print('   %-70s calls %5s avg %8.1f us total %8.2f ms' % (r['Name'].replace('void ','').replace('wsk::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
python3 -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
