#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for v in "pl_warm_disp_pct=100" "pl_warm_disp_pct=50" "pl_warm_disp_pct=25" "pl_warm_disp_pct=0" "pl_warm_disp_pct=100"; do
  timeout 400 python tools/pos_probe.py c3 - 3 $v 2>>$O/err.txt | tail -1
done | tee $O/pos.txt
