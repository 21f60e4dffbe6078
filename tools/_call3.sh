#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for c in c3; do for v in "pl_hard_cand=1500" "pl_hard_cand=3000" "pl_hard_cand=5000" "pl_hard_cand=0" "pl_hard_cand=1500" "pl_hard_cand=3000"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v" timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>$O/${c}_$n.err | grep '^{"metric"' > $O/${c}_$n.json
  python - <<PY
import json
d=json.load(open("$O/${c}_$n.json"))
print("$c $v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done; done
