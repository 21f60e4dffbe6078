#!/bin/bash
mkdir -p gpurun_out/r2an; export TMPDIR=/tmp
O=gpurun_out/r2an
for c in c2 c3 c5; do timeout 170 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2an/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1], round(d["value"], 1), round(d["ms_per_step"], 4))
    except Exception as e:
        print(f, "unreadable", e)
PY
