#!/bin/bash
mkdir -p gpurun_out/r2ad; export TMPDIR=/tmp
O=gpurun_out/r2ad
run() { MP2P_HIP_TUNE="$2" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run a_d5 "" "--defer 5"
run a_d6 "" "--defer 6"
run a_d8 "" "--defer 8"
run a_d8_cap "tile_cand_cap=20000,tile_time_cap_us=100" "--defer 8"
run b_d6 "" "--scene b --defer 6"
run b_d8 "" "--scene b --defer 8"
run b_d12 "" "--scene b --defer 12"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2ad/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, round(d["nn_stats"]["deferred_to_one_query_kernel_frac"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
