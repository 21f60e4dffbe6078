#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_FUZZ_HORN_SEEDS=0:1500 timeout 1500 python -m pytest tests/test_gpu_horn.py tests/test_gpu_icp.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider -k "(horn or icp or c2) and not c5" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
for c in c2 c2 c2; do
  timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>$O/$c.err | grep '^{"metric"' > $O/$c.json
  python - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"])
PY
done | tee $O/c.txt
