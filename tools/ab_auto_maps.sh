#!/bin/bash
# The four smooth 8192^2 maps with the PRODUCT library as shipped (persistent pass: auto) and with the passes forced (tuning build, WS_RELAX_PERSIST=0)
set -o pipefail
tag=${1:-aam}; out=gpurun_out/$tag; mkdir -p $out
for c in 4 16 64 256; do
  echo "== corr $c product (auto)" >> $out/ab.txt
  timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; exit 1; }
done
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in 64 256; do
  echo "== corr $c tuning build, flood order forced" >> $out/ab.txt
  WS_RELAX_PERSIST=2 WS_RELAX_PERSIST_DIAG=1 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; exit 1; }
  grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
done
cat $out/ab.txt
