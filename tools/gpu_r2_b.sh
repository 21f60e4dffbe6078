#!/bin/bash
# round 2, call B: segmented query lists; knob A/B; rocprof kernel stats
mkdir -p gpurun_out/r2b; export TMPDIR=/tmp
O=gpurun_out/r2b
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
run() { # name, tune, extra args
  MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run default ""
run lane0 "lane_cells=0"
run lane2 "lane_cells=2"
run lane3 "lane_cells=3"
run nodedup "claim_dedup=0,claim_peek=0"
run cap2k "tile_cand_cap=2048"
run cap1k3 "tile_cand_cap=1024,lane_cells=3"
run cold "" "--cold"
run q16 "" "--q 16"
run q64 "" "--q 64"
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/$O/prof_kt.log 2>&1; echo "rocprof rc=$?"
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2b/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"},
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["nn_stats"].items() if k.startswith("lane")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "chain step" $O/bench_default.err | head -4
find $O/prof_kt -name "*kernel_stats*" | head -3
f=$(find $O/prof_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
