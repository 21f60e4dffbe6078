#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_FUZZ_HORN_SEEDS=0:600 timeout 1500 python -m pytest tests/test_gpu_horn.py tests/test_gpu_icp.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_matcher_inlier_ratio.py tests/test_gpu_gn.py -x -q -m gpu -p no:cacheprovider -k "(horn or icp or c2 or adaptive or inlier or covariance or quality) and not c5" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -3 $O/pytest.log
for c in c2 c2; do
  timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>$O/$c.err | grep '^{"metric"' > $O/$c.json
  python - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done | tee $O/c.txt
