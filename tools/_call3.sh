#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python tools/ab_probe.py $O/ab.json "f0:far_pass=0" "f1:far_pass=1" "f0b:far_pass=0" "f1b:far_pass=1" > $O/ab.txt 2> $O/ab.err; tail -4 $O/ab.txt | cut -c1-200
for v in "far_pass=0" "far_pass=1"; do
  MP2P_HIP_TUNE="$v" timeout 600 python bench.py --config c2 --steps 40 --warmup 5 2>$O/c2.err | grep '^{"metric"' > $O/c2_$v.json
  python - <<PY
import json
d=json.load(open("$O/c2_$v.json"))
print("c2 $v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"])
PY
done
