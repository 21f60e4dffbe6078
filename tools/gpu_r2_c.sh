#!/bin/bash
# round 2, call C: parameter sweep with the lane kernel as prologue only + timeline of the tile kernel
mkdir -p gpurun_out/r2c; export TMPDIR=/tmp
O=gpurun_out/r2c
run() { # name, tune, extra args
  MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run default ""
run q16 "" "--q 16"
run q16d2 "" "--q 16 --defer 2"
run q16d3 "" "--q 16 --defer 3"
run q16d6 "" "--q 16 --defer 6"
run q16g15 "" "--q 16 --grp 1.5"
run q16g4 "" "--q 16 --grp 4"
run q16cap1k "tile_cand_cap=1024" "--q 16"
run q16cap512 "tile_cand_cap=512" "--q 16"
run q16b32 "" "--q 16 --bricks 32"
run q16b512 "" "--q 16 --bricks 512"
run q32d2 "" "--defer 2"
timeout 200 python tools/timeline_probe.py > $O/timeline.log 2>&1; echo "timeline rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
cut -c1-1500 $O/timeline.log
