#!/bin/bash
mkdir -p gpurun_out/r2w; export TMPDIR=/tmp
O=gpurun_out/r2w
run() { MP2P_HIP_TUNE="$2" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run a_tpc3 "" "--target-per-cell 3"
run a_tpc12 "" "--target-per-cell 12"
run b_tpc3 "" "--scene b --target-per-cell 3"
run b_tpc12 "" "--scene b --target-per-cell 12"
run b_tpc24 "" "--scene b --target-per-cell 24"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2w/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, d["nn_stats"]["voxel_m"], round(d["nn_stats"]["candidates_tested_per_query"], 1), round(d["nn_stats"]["deferred_to_one_query_kernel_frac"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
