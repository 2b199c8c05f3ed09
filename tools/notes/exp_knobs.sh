#!/bin/bash
# A/B runs of tools/exp_one.py under the -DWS_TUNING build.  usage: exp_knobs.sh <outfile> <workload> <N> "<ENV1=..> <ENV2=..>" ...
out=$1; wl=$2; n=$3; shift 3
export WS_HIP_LIB=$GRAFT_REPO_ROOT/rustronomy-watershed_amd/libws_hip_tuning.so
for setting in "$@"; do
  env $setting python tools/exp_one.py $wl $n >> $out 2>&1 || echo "FAILED: $setting" >> $out
done
