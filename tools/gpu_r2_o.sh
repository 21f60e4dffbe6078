#!/bin/bash
mkdir -p gpurun_out/r2o; export TMPDIR=/tmp
O=gpurun_out/r2o
timeout 1200 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_matcher_pt2pl.py tests/test_gpu_gn.py tests/test_gpu_icp.py tests/test_gpu_split_phases.py tests/test_gpu_comm.py tests/test_gpu_fullsize_properties.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_matcher_inlier_ratio.py -q -x --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() { MP2P_HIP_TUNE="$2" timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run dir ""
run nodir "dir_budget_mb=0"
run dir2 ""
run dir_sceneb "" "--scene b"
run nodir_sceneb "dir_budget_mb=0" "--scene b"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2o/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), d["step_ms"], {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -h "timed steps\|index" $O/bench_dir.err $O/bench_dir_sceneb.err
