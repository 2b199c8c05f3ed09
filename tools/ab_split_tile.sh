#!/bin/bash
# 128 x 64 against 128 x 128 tiles in the same-grid passes / the queue (a build with -DWS_SPLIT_NW=16: libws_hip_nw16.so), four smooth 8192^2 maps
set -o pipefail
tag=${1:-abs}; out=gpurun_out/$tag; mkdir -p $out
for lib in libws_hip_tuning.so libws_hip_nw16.so; do
  export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/$lib
  for c in 4 16 64 256; do
    for m in 0 2; do
      echo "== $lib corr $c persist $m" >> $out/ab.txt
      WS_RELAX_PERSIST=$m WS_RELAX_PERSIST_DIAG=1 timeout -k 10 120 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; tail -3 $out/diag.txt >> $out/ab.txt; exit 1; }
      grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
