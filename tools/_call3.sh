#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pl.py tests/test_gpu_comm.py tests/test_gpu_bench_two_ranks.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
for c in c5 c3; do
  timeout 600 python bench.py --config $c --steps 30 --warmup 5 2>$O/$c.err | grep '^{"metric"' > $O/$c.json
  python - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", round(d["value"],1), round(d["ms_per_step"],4), d["step_ms"], {k:round(v,4) for k,v in d["kernel_ms"].items() if isinstance(v,float)})
PY
done | tee $O/c.txt
MP2P_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/dist.err | grep '^{"metric"' > $O/dist.json
python - <<PY
import json
d=json.load(open("$O/dist.json"))
print("default line, sharded step on a one-rank communicator:", round(d["value"],1), round(d["ms_per_step"],4), d.get("step_ms"))
PY
