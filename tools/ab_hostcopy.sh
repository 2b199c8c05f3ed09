#!/bin/bash
# gpurun: tools/ab_hostcopy.sh -- the chunked host copy (csrc/ws_hostcopy.hip) under thread counts and chunk sizes; the knobs are
# environment variables of the TUNING build only (the product takes ws_ctx_set_host_threads)
python -c "import __graft_entry__ as g; g.build_hip(tuning=True)" || exit 1
export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for k in 21 22 23; do for t in 2 3 4 6 8; do WS_HOST_THREADS=$t WS_HOST_CHUNK_LOG2=$k python tools/exp_hostcopy.py || exit 1; done; done
WS_HOST_THREADS=0 python tools/exp_hostcopy.py
WS_HOST_THREADS_FAIL=1 python tools/exp_hostcopy.py      # threads cannot start: the calling thread widens chunk after chunk
nproc; lscpu | grep -i "model name\|socket\|numa node(s)"
