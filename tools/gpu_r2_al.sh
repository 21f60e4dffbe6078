#!/bin/bash
mkdir -p gpurun_out/r2al; export TMPDIR=/tmp
O=gpurun_out/r2al
run() { MP2P_HIP_TUNE="$2" timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run a_w5 "single_waves=5"
run a_w4 ""
run a_w5b40 "single_waves=5,single_blocks_per_cu=40"
run b_w5 "single_waves=5" "--scene b"
run b_w4 "" "--scene b"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2al/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
