#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_horn.py tests/test_gpu_icp.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "not c5 and not c4 and not c3" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
for c in c2 c2; do
  timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>$O/$c.err | grep '^{"metric"' > $O/$c.json
  python - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done | tee $O/c.txt
