#!/bin/bash
# launch by launch, one smooth-map transform: tools/trace_smooth.sh <N> <corr> <tag> [persistent_pass]  ->  gpurun_out/<tag>/smooth_<corr>_launches.txt
n=${1:-8192}; corr=${2:-64}; tag=${3:-smooth}; pp=${4:-0}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/kts -- python3 $root/tools/exp_smooth_trace.py $n $corr 1 $pp > $out/smooth_$corr.log 2>&1 || exit 1
f=$(find $out/kts -name "*kernel_trace.csv" | head -1)
python3 $root/tools/trace_last_transform.py $f > $out/smooth_${corr}_launches.txt
rm -rf $out/kts
head -40 $out/smooth_${corr}_launches.txt
