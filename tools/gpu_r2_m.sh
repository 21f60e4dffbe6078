#!/bin/bash
mkdir -p gpurun_out/r2m; export TMPDIR=/tmp
O=gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_comm.py -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
run() { MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run default ""
run w6 "single_waves=6"
run w8 "single_waves=8"
run w8b64 "single_waves=8,single_blocks_per_cu=64"
run w6b48 "single_waves=6,single_blocks_per_cu=48"
run b20 "single_blocks_per_cu=20"
run b64 "single_blocks_per_cu=64"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2m/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
