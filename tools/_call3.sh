#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_FUZZ_SEEDS=48:448 MP2P_FUZZ_PT_SEEDS=36:1036 MP2P_FUZZ_PL_SEEDS=14:1014 MP2P_FUZZ_PLSEQ_SEEDS=1500:2300 MP2P_FUZZ_GN_SEEDS=0:6000 MP2P_FUZZ_HORN_SEEDS=0:2000 MP2P_FUZZ_LAYER_SEEDS=0:300 MP2P_FUZZ_HOST_SEEDS=0:200 MP2P_FUZZ_DECIM_SEEDS=0:300 \
  timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -x > $O/fuzz.log 2>&1; echo "fuzz rc=$? $(grep -E 'passed|failed' $O/fuzz.log | tail -1)" | tee $O/rc.txt
tail -4 $O/fuzz.log
