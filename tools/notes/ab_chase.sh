#!/bin/bash
# A/B of k_resolve_chase builds (libws_hip_<name>.so beside the package) under rocprofv3 --kernel-trace: usage ab_chase.sh name...
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/chab; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  export WS_HIP_LIB=$root/rustronomy-watershed_amd/libws_hip_$L.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$L -- python3 $root/tools/exp_one.py ${WL:-noise} 8192 > $out/$L.log 2>&1 || exit 1
  f=$(find $out/kt_$L -name "*kernel_stats.csv" | head -1); echo "== $L" >> $out/sum.txt; grep "8192:" $out/$L.log >> $out/sum.txt
  python3 - $f >> $out/sum.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_resolve_chase','k_seed_tables','k_resolve_local','k_relax')):
        print('   %-60s calls %4s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf $out/kt_$L
done
cat $out/sum.txt
