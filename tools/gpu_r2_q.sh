#!/bin/bash
mkdir -p gpurun_out/r2q; export TMPDIR=/tmp
O=gpurun_out/r2q
run() { MP2P_HIP_TUNE="$2" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run keep0 "keep_rule=0"
run keep1 ""
run keep0b "keep_rule=0"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2q/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, {k: round(v, 3) for k, v in d["nn_stats"].items() if "frac" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
