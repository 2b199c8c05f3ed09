#!/bin/bash
# gpurun: tools/trace_small.sh [size] -- launch by launch, one device-resident transform of a small plane (default 2048^2)
S=${1:-2048}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_small && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/trace_small
cat > /tmp/one_small.py <<PY
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
img = eng.random_field($S, $S, 1)
seeds = eng.find_local_minima(img)
out = torch.empty(($S, $S), dtype=torch.int32, device=img.device)
for i in range(8):
    eng.segment(img, seeds, out=out); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_small -o t -- python3 /tmp/one_small.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/trace_small -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_last_transform.py $f
