#!/bin/bash
mkdir -p gpurun_out/r2ak; export TMPDIR=/tmp
O=gpurun_out/r2ak
run() { MP2P_HIP_TUNE="$2" timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"; }
run base ""
run h50 "hard_radius_pct=50"
run h200 "hard_radius_pct=200"
run cap3k "tile_cand_cap=3072"
run cap12k "tile_cand_cap=12288"
run grp2 "" "--grp 2.0"
run grp35 "" "--grp 3.5"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2ak/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), round(d["step_ms"]["median"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"}, round(d["nn_stats"]["deferred_to_one_query_kernel_frac"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
