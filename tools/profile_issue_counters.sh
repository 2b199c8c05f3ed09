#!/bin/bash
# SQ issue / wait counters of the transform's kernels, in --pmc passes of their own:
#   tools/profile_issue_counters.sh <tag>  ->  gpurun_out/<tag>/issue_counters.json (+ the counter list of the box)
set -u
tag=${1:-issue}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counter_list.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pa -- python3 $root/tools/pmc_probe.py --steps 3 > $out/pa.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pb -- python3 $root/tools/pmc_probe.py --steps 3 > $out/pb.log 2>&1
fa=$(find $out/pa -name "*counter_collection.csv" | head -1); fb=$(find $out/pb -name "*counter_collection.csv" | head -1)
python3 $root/tools/pmc_counters_summary.py $fa $fb > $out/issue_counters.json 2> $out/summary.err
rm -rf $out/pa $out/pb
ls -la $out
