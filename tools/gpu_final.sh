#!/bin/bash
# round-end evidence in one bounded call: GPU tests, the bench line, the rocprofv3 kernel stats
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
timeout 40 python -m pytest tests -m gpu -q -x --timeout=60 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 35 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_big.json 2> gpurun_out/bench_big.err; echo "bench rc=$?"
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/bench_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof/bench_kt.log 2>&1; echo "rocprof rc=$?"
cd $R
python - <<'PY'
import json
for f in ("gpurun_out/bench_big.json", "gpurun_out/bench_noorder.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
