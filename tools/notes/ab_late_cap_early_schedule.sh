export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for cap in 2 3 4 5; do for c in 8 16; do echo "== late cap $cap corr $c"; WS_RELAX_LATE_CAP=$cap timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 2>/dev/null | cut -c1-100; done; done
