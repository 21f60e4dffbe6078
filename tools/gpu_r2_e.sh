#!/bin/bash
# round 2, call E: hard tiles first; GN agent-atomics variant; spin wait
mkdir -p gpurun_out/r2e; export TMPDIR=/tmp
O=gpurun_out/r2e
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
run() { # name, tune, extra args
  MP2P_HIP_TUNE="$2" timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run default ""
run alleasy "hard_radius_pct=10000000"
run hard60 "hard_radius_pct=60"
run hard150 "hard_radius_pct=150"
run hard250 "hard_radius_pct=250"
run gn2 "gn_ticket=2"
run nospin "sync_spin=0"
run hard100nocap "tile_time_cap_us=1000000"
MP2P_HIP_TUNE="" timeout 200 python tools/timeline_probe.py > $O/timeline.log 2>&1; echo "timeline rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2e/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
    except Exception as e:
        print(f, "unreadable", e)
PY
cut -c1-900 $O/timeline.log
