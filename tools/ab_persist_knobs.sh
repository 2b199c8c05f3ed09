#!/bin/bash
# Queue in flood order (WS_RELAX_PERSIST=2) under poll-rate and worker-count knobs, one smooth 8192^2 map.  Tuning build.
set -o pipefail
tag=${1:-abk}; corr=${2:-64}
out=gpurun_out/$tag; mkdir -p $out
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
run() { echo "== $*" >> $out/ab.txt; env "$@" WS_RELAX_PERSIST=2 WS_RELAX_PERSIST_DIAG=1 timeout -k 10 120 python tools/exp_one.py smooth$corr 8192 3 >> $out/ab.txt 2>$out/diag.txt || exit 1; grep "persistent pass\|per tile run" $out/diag.txt | tail -2 >> $out/ab.txt; }
for w in 1024 512 256 128; do run WS_RELAX_PERSIST_WORKERS=$w; done
run WS_RELAX_PERSIST_WORKERS=256 WS_RELAX_PERSIST_CAP=12
run WS_RELAX_PERSIST_WORKERS=256 WS_RELAX_PERSIST_CAP=3
cat $out/ab.txt
