#!/bin/bash
# round 2, call A: parity of the restructured search + A/B of its knobs (MP2P_HIP_TUNE)
mkdir -p gpurun_out/r2a; export TMPDIR=/tmp
O=gpurun_out/r2a
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
run() { # name, tune, extra args
  MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run default ""
run lane0 "lane_cells=0"
run lane2 "lane_cells=2"
run lane3 "lane_cells=3"
run nodedup "claim_dedup=0,claim_peek=0"
run nopeek "claim_peek=0"
run cap2k "tile_cand_cap=2048"
run nocap "tile_cand_cap=100000000"
run cold "" "--cold"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2a/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"},
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["nn_stats"].items() if k.startswith("lane")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "chain step" $O/bench_default.err | head -12
