#!/bin/bash
# flood-order queue forced (tuning build, WS_RELAX_PERSIST=2) against the passes (=0) on the four smooth 8192^2 maps
set -o pipefail
tag=${1:-aqf}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in 4 16 64 256; do
  for m in 0 2; do
    echo "== corr $c persist $m" >> $out/ab.txt
    WS_RELAX_PERSIST=$m WS_RELAX_PERSIST_DIAG=1 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; exit 1; }
    grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
  done
done
cat $out/ab.txt
