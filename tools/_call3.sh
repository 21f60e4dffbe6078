#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for v in 1000 1500 2000 3000; do
  MP2P_HIP_TUNE="pl_hard_cand=$v" timeout 600 python bench.py --config c3 --steps 40 --warmup 5 2>$O/c3.err | grep '^{"metric"' > $O/c3_${v}_$rep.json
  python - <<PY
import json
d=json.load(open("$O/c3_${v}_$rep.json"))
print("c3 pl_hard_cand=$v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], round(d["kernel_ms"]["search_pt2pl"],4))
PY
done; done | tee $O/c.txt
