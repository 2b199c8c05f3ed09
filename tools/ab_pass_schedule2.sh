#!/bin/bash
# the early pass schedule (same grid from 3, scans from 2, lists from 3) against the shipped one on denser maps; passes forced
set -o pipefail
tag=${1:-aps2}; out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for c in 5 6 8 10; do
  for cfg in "7 4 6" "3 2 3"; do
    set -- $cfg
    echo "== same_from $1 scan_from $2 list_from $3 corr $c" >> $out/ab.txt
    WS_RELAX_PERSIST=0 WS_RELAX_SAME_GRID_FROM=$1 WS_RELAX_SCAN_FROM=$2 WS_RELAX_LIST_FROM=$3 timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 >> $out/ab.txt 2>$out/diag.txt || { echo FAILED >> $out/ab.txt; tail -3 $out/diag.txt >> $out/ab.txt; }
  done
done
echo "== noise 7 4 6" >> $out/ab.txt; WS_RELAX_PERSIST=0 timeout -k 10 300 python tools/exp_one.py noise 8192 5 >> $out/ab.txt 2>/dev/null
echo "== noise 3 2 3" >> $out/ab.txt; WS_RELAX_PERSIST=0 WS_RELAX_SAME_GRID_FROM=3 WS_RELAX_SCAN_FROM=2 WS_RELAX_LIST_FROM=3 timeout -k 10 300 python tools/exp_one.py noise 8192 5 >> $out/ab.txt 2>/dev/null
cat $out/ab.txt
