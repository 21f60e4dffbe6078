#!/bin/bash
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/nn_one.py gt 1 > /dev/null 2>&1
cd /tmp
for pose in gt init; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof/p1_$pose -o p -- python $R/tools/nn_one.py $pose 2 > $R/gpurun_out/prof/p1_$pose.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof/p2_$pose -o p -- python $R/tools/nn_one.py $pose 2 > $R/gpurun_out/prof/p2_$pose.log 2>&1
done
cd $R
