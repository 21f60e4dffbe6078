#!/bin/bash
mkdir -p gpurun_out/r2am; export TMPDIR=/tmp
O=gpurun_out/r2am
timeout 600 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_properties.py tests/test_gpu_split_phases.py tests/test_gpu_icp.py -q -x --timeout=300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
run() { MP2P_HIP_TUNE="$2" timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run a_def ""
run a_b60 "single_blocks_per_cu=60"
run a_b20 "single_blocks_per_cu=20"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2am/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
