#!/bin/bash
# round 2, call I: the whole GPU suite, the default bench line with its extra blocks, the config lines
mkdir -p gpurun_out/r2i; export TMPDIR=/tmp
O=gpurun_out/r2i
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
for c in c2 c3 c5; do timeout 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2i/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), d["step_ms"], {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
print("roofline", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k != "kernel"})
print("stability", d.get("stability"))
print("host_boundary", d.get("host_boundary"))
sb = d.get("scene_b", {})
print("scene_b", {k: v for k, v in sb.items() if k not in ("roofline",)})
print("scene_b roofline", sb.get("roofline"))
print("cpu", d.get("cpu_baseline"))
for c in ("c2", "c3", "c5"):
    try:
        e = json.loads(open(f"gpurun_out/r2i/bench_{c}.json").read().strip().splitlines()[-1])
        print(c, round(e["value"], 1), round(e["ms_per_step"], 3), e["kernel_ms"], e["pairs_per_step"], e["final_pose_error"], round(e["roofline"]["frac"], 5))
    except Exception as ex:
        print(c, "unreadable", ex)
PY
tail -3 $O/bench_default.err | cut -c1-300
