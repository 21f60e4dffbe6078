#!/bin/bash
# round 2, call D: fused Gauss-Newton iteration (ticket), fused compaction (bbox), tile time cap
mkdir -p gpurun_out/r2d; export TMPDIR=/tmp
O=gpurun_out/r2d
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
run() { # name, tune, extra args
  MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run default ""
run noticket "gn_ticket=0"
run nofuse "compact_fused=0"
run cap20 "tile_time_cap_us=20"
run cap25 "tile_time_cap_us=25"
run cap50 "tile_time_cap_us=50"
run nocap "tile_time_cap_us=1000000"
run cap25c2k "tile_time_cap_us=25,tile_cand_cap=2048"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2d/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
