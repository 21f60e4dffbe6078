#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_icp.py tests/test_gpu_comm.py tests/test_gpu_matcher_pt2pt.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_multilayer.py tests/test_gpu_boundary_hostpath.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -3 $O/pytest.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/b.err | grep '^{"metric"' > $O/b$i.json
python - <<PY
import json
d=json.load(open("$O/b$i.json"))
print("default", round(d["value"],1), round(d["ms_per_step"],4), d.get("step_ms"), d["kernel_ms"]["step_minus_kernels"], d["stability"]["iterations_per_s"])
PY
done
