#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for v in "persist=0" "persist=1,heavy_cand=0" "persist=1"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v,nn_cert=0" timeout 400 python tools/timeline_probe.py 1000000 10000000 b > $O/tl_$n.json 2>$O/tl_$n.err
  echo "== $v"; grep '"pose": "chain"' $O/tl_$n.json | cut -c1-900
done
