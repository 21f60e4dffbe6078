#!/bin/bash
mkdir -p gpurun_out/r2s; export TMPDIR=/tmp
O=gpurun_out/r2s
run() { MP2P_HIP_TUNE="$2" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run base ""
run q64 "" "--q 64"
run q16 "" "--q 16"
run cap25 "tile_time_cap_us=25"
run cap100 "tile_time_cap_us=100,tile_cand_cap=20000"
run b_base "" "--scene b"
run b_q64 "" "--scene b --q 64"
run b_cap25 "tile_time_cap_us=25" "--scene b"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, {k: round(v, 3) for k, v in d["nn_stats"].items() if "frac" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
