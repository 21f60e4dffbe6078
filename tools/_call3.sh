#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_FUZZ_PLSEQ_SEEDS=0:400 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_matcher_pt2pl.py -x -q -m gpu -p no:cacheprovider -k "pose_seq or pt2pl" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -3 $O/pytest.log
for c in c3 c5; do for v in "pl_cert_step_mm=10" "pl_cert_step_mm=0"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v" timeout 600 python bench.py --config $c --steps 40 --warmup 5 2>$O/${c}_$n.err | grep '^{"metric"' > $O/${c}_$n.json
  python - <<PY
import json
d=json.load(open("$O/${c}_$n.json"))
print("$c $v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done; done
