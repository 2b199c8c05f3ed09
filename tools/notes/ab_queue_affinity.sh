#!/bin/bash
# the queue in flood order with and without XCD affinity (WS_RELAX_PERSIST_AFFINITY, tuning build), sparse-seed smooth 8192^2 maps
set -o pipefail
tag=${1:-aqa}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in 64 256 32; do
  for a in 0 1; do
    echo "== corr $c affinity $a" >> $out/ab.txt
    WS_RELAX_PERSIST=2 WS_RELAX_PERSIST_AFFINITY=$a WS_RELAX_PERSIST_DIAG=1 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; tail -3 $out/diag.txt >> $out/ab.txt; exit 1; }
    grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
  done
done
cat $out/ab.txt
