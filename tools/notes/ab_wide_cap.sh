export WS_HIP_LIB=$PWD/rustronomy-watershed_amd/libws_hip_tuning.so
for wc in "0,0" "3,6" "6,6" "3,9" "6,4" "10,5"; do for c in 8 16; do echo "== wide cap $wc corr $c"; WS_RELAX_WIDE_CAP=$wc timeout -k 10 300 python tools/exp_one.py smooth$c 8192 3 2>/dev/null | cut -c1-100; done; done
