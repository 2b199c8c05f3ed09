#!/bin/bash
# rounds per tile run in the late passes (WS_RELAX_LATE_CAP, tuning build), passes forced, four smooth 8192^2 maps
set -o pipefail
tag=${1:-alc}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for cap in 3 4 6 8; do
  for c in 4 16 64; do
    echo "== cap $cap corr $c" >> $out/ab.txt
    WS_RELAX_PERSIST=0 WS_RELAX_LATE_CAP=$cap timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; exit 1; }
  done
done
cat $out/ab.txt
