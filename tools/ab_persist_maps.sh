#!/bin/bash
# Smooth 8192^2 maps, correlation 4 / 16 / 64 / 256 px: ordinary passes against the queue in flood order with W workers.  Tuning build.
# usage (gpurun): tools/ab_persist_maps.sh <tag> [workers ...]
set -o pipefail
tag=${1:-abm}; shift
ws=${@:-256}
out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in 4 16 64 256; do
  echo "== corr $c passes" >> $out/ab.txt
  WS_RELAX_PERSIST=0 timeout -k 10 120 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || exit 1
  for w in $ws; do
    echo "== corr $c flood order, $w workers" >> $out/ab.txt
    WS_RELAX_PERSIST=2 WS_RELAX_PERSIST_WORKERS=$w WS_RELAX_PERSIST_DIAG=1 WS_RELAX_PERSIST_MODE=${PMODE:-0} timeout -k 10 120 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || exit 1
    grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
  done
done
cat $out/ab.txt
