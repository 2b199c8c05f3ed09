#!/bin/bash
# Persistent pass A/B on smooth 8192^2 maps: ordinary passes, first-come queue (1), flood-order buckets (2).  Needs the tuning build.
# usage (gpurun): tools/ab_persist_order.sh <tag> [corr ...]
set -o pipefail
tag=${1:-ab}; shift
corrs=${@:-4 16 64 256}
out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in $corrs; do
  for m in 0 1 2; do
    echo "== corr $c persist $m" >> $out/ab.txt
    WS_RELAX_PERSIST=$m WS_RELAX_PERSIST_DIAG=1 timeout -k 10 120 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag_${c}_$m.txt || { echo "FAILED corr $c mode $m" >> $out/ab.txt; tail -5 $out/diag_${c}_$m.txt >> $out/ab.txt; exit 1; }
    grep "persistent pass\|per tile run" $out/diag_${c}_$m.txt | tail -2 >> $out/ab.txt
  done
done
cat $out/ab.txt
