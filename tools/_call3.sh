#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for v in "pl_select=-1" "pl_select=1" "pl_select=1,pl_sel_hard_cand=1000" "pl_select=1,pl_sel_hard_cand=300" "pl_select=1,pl_sel_hard_cand=1" "pl_select=1,pl_waves=1"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v" timeout 400 python bench.py --config c3 --steps 40 --warmup 5 2>$O/c3_$n.err | grep '^{"metric"' > $O/c3_$n.json
  python - <<PY
import json
d=json.load(open("$O/c3_$n.json"))
print("$v", round(d["value"]), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done
