#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
MP2P_HIP_TUNE=pl_select=1 MP2P_FUZZ_PLSEQ_SEEDS=0:600 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_matcher_pt2pl.py -x -q -m gpu -p no:cacheprovider -k "pose_seq or pt2pl" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -3 $O/pytest.log
for v in "pl_empty_room=0" "pl_empty_room=1" "pl_empty_room=0" "pl_empty_room=1"; do
  n=$(echo $v | tr ',=' '__')
  MP2P_HIP_TUNE="$v" timeout 600 python bench.py --config c5 --steps 30 --warmup 5 2>$O/c5.err | grep '^{"metric"' > $O/c5_$n.json
  python - <<PY
import json
d=json.load(open("$O/c5_$n.json"))
print("c5 $v", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], round(d["kernel_ms"]["search_pt2pl"],4))
PY
done | tee $O/c.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "c5 and Cauchy" > $O/pytest2.log 2>&1; echo "c5 test rc=$? $(grep -E 'passed|failed' $O/pytest2.log | tail -1)" | tee -a $O/rc.txt
