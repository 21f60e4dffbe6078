#!/bin/bash
# the bench lines kept under profiles/: default command, scene B, the other BASELINE configs
mkdir -p gpurun_out/r2ac; export TMPDIR=/tmp
O=gpurun_out/r2ac
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "default rc=$?"
for c in c2 c3 c5; do timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2ac/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), json.dumps(d.get("roofline"))[:300], json.dumps(d.get("kernel_ms"))[:300])
    except Exception as e:
        print(f, "unreadable", e)
PY
