#!/bin/bash
# The merging transform at 8192^2 (BASELINE config 3), final labels and transform_to_list: kernel stats and HBM bytes
# (FETCH_SIZE / WRITE_SIZE in --pmc passes of their own, calibrated in the same run):
#   tools/profile_merge.sh <tag>  ->  gpurun_out/<tag>/{merge,tolist}_{kernel_stats.csv,pmc_hbm_bytes.json}
set -u
tag=${1:-mprof}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
for form in merge tolist; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/tools/pmc_probe.py --steps 3 --$form > $out/${form}_kt.log 2>&1 || exit 1
  f=$(find $out/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${form}_kernel_stats.csv; rm -rf $out/kt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $root/tools/pmc_probe.py --steps 2 --$form > $out/${form}_pf.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -- python3 $root/tools/pmc_probe.py --steps 2 --$form > $out/${form}_pw.log 2>&1 || exit 1
  ff=$(find $out/pf -name "*counter_collection.csv" | head -1); fw=$(find $out/pw -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_summarise.py $ff $fw $((8192*8192)) > $out/${form}_pmc_hbm_bytes.json 2> $out/${form}_pmc.err
  rm -rf $out/pf $out/pw
done
ls -la $out
