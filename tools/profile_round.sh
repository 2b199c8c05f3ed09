#!/bin/bash
# Collects a round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>   ->  gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_hbm_bytes.json, tolist_kernel_stats.csv, smooth_kernel_stats.csv}
# Kernel traces and PMC counters in runs of their own (never combined); the program itself after `--`.
set -u
tag=${1:-prof}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --steps 5 --warmup 3 --no-extras --no-pipeline > $out/kt.log 2>&1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/tools/pmc_probe.py --steps 3 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/tools/pmc_probe.py --steps 3 > $out/pmc_write.log 2>&1
ff=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1); fw=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 $root/tools/pmc_summarise.py $ff $fw $((8192*8192)) > $out/pmc_hbm_bytes.json 2> $out/pmc_sum.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_tolist -- python3 $root/tools/exp_tolist_raw.py 1024 > $out/tolist.log 2>&1
f=$(find $out/kt_tolist -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/tolist_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_smooth -- python3 $root/tools/exp_smooth_trace.py 8192 64 > $out/smooth.log 2>&1
f=$(find $out/kt_smooth -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/smooth_kernel_stats.csv
rm -rf $out/kt $out/pmc_fetch $out/pmc_write $out/kt_tolist $out/kt_smooth
ls -la $out
