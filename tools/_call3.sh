#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for v in "pl_hq=4,pl_hard_cand=3000" "pl_hq=8,pl_hard_cand=600" "pl_hq=8,pl_hard_cand=1000" "pl_hq=8,pl_hard_cand=1500" "pl_hq=8,pl_hard_cand=3000" "pl_hq=4,pl_hard_cand=3000" "pl_hq=8,pl_hard_cand=1000"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v" timeout 600 python bench.py --config c3 --steps 40 --warmup 5 2>$O/c3.err | grep '^{"metric"' > $O/c3_$n.json
  python - <<PY
import json
d=json.load(open("$O/c3_$n.json"))
print("c3 $v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], round(d["kernel_ms"]["search_pt2pl"],4))
PY
done | tee $O/c.txt
for v in "pl_hq=8,pl_hard_cand=1000" "pl_hq=8,pl_hard_cand=600"; do echo "== $v"; MP2P_HIP_TUNE=$v timeout 600 python tools/pl_timeline.py 120000 1 2>>$O/err.txt | cut -c1-700; done | tee $O/pltl.txt
