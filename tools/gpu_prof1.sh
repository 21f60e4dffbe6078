#!/bin/bash
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
python tools/nn_one.py gt 1 > /dev/null 2>&1   # warm the input cache
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt_gt -o kt -- python $GRAFT_REPO_ROOT/tools/nn_one.py gt 5 > $GRAFT_REPO_ROOT/gpurun_out/prof/kt_gt.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt_init -o kt -- python $GRAFT_REPO_ROOT/tools/nn_one.py init 5 > $GRAFT_REPO_ROOT/gpurun_out/prof/kt_init.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/nn_one.py gt 2 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/nn_one.py gt 2 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*.csv" | head -20
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo == $f; head -12 $f | cut -c1-200; done
