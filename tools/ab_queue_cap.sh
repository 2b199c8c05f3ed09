#!/bin/bash
# rounds per tile run of the flood-order queue (WS_RELAX_PERSIST_CAP, tuning build) on the two sparse-seed maps
set -o pipefail
tag=${1:-aqc}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for cap in 3 6 12 24; do
  for c in 64 256; do
    echo "== cap $cap corr $c" >> $out/ab.txt
    WS_RELAX_PERSIST=2 WS_RELAX_PERSIST_CAP=$cap WS_RELAX_PERSIST_DIAG=1 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; exit 1; }
    grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
  done
done
cat $out/ab.txt
