#!/bin/bash
# pass-schedule knobs (tuning build, passes forced) on the wide smooth maps: WS_RELAX_SAME_GRID_FROM x WS_RELAX_SCAN_FROM x WS_RELAX_LIST_FROM
set -o pipefail
tag=${1:-aps}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for cfg in "7 4 6" "5 4 4" "5 3 4" "3 2 3" "7 3 6" "7 2 6" "9 4 6"; do
  set -- $cfg
  for c in 4 12 16; do
    echo "== same_from $1 scan_from $2 list_from $3 corr $c" >> $out/ab.txt
    WS_RELAX_PERSIST=0 WS_RELAX_SAME_GRID_FROM=$1 WS_RELAX_SCAN_FROM=$2 WS_RELAX_LIST_FROM=$3 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; tail -3 $out/diag.txt >> $out/ab.txt; }
  done
done
cat $out/ab.txt
