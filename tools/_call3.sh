#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_FUZZ_PLSEQ_SEEDS=0:200 timeout 900 python -m pytest tests/test_gpu_matcher_pt2pl.py tests/test_gpu_matcher_inlier_ratio.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider -k "pt2pl or inlier or adaptive or pose_seq" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
for c in c5 c3 c5 c3; do
  timeout 600 python bench.py --config $c --steps 30 --warmup 5 2>$O/$c.err | grep '^{"metric"' > $O/$c.json
  python - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done | tee $O/c.txt
