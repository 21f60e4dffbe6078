#!/bin/bash
mkdir -p gpurun_out/r2x; export TMPDIR=/tmp
O=gpurun_out/r2x
run() { MP2P_HIP_TUNE="$2" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run a_f_d8 "" "--target-per-cell 1.5 --defer 8"
run a_f_d4 "" "--target-per-cell 1.5 --defer 4"
run a_f_d8_h200 "hard_radius_pct=200" "--target-per-cell 1.5 --defer 8"
run b_f_d12 "" "--scene b --target-per-cell 3 --defer 12"
run b_f_d8 "" "--scene b --target-per-cell 3 --defer 8"
run b_f_d12_h200 "hard_radius_pct=200" "--scene b --target-per-cell 3 --defer 12"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2x/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, round(d["nn_stats"]["voxel_m"], 3), round(d["nn_stats"]["candidates_tested_per_query"], 1), round(d["nn_stats"]["deferred_to_one_query_kernel_frac"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
