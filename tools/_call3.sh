#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for v in "pl_sel_hard_large=1500" "pl_sel_hard_large=400"; do
echo "== $v"; MP2P_HIP_TUNE="$v" timeout 600 python tools/pl_timeline.py 1000000 2>$O/pltl.err | tee -a $O/pltl.txt | cut -c1-1700
done
