#!/bin/bash
mkdir -p gpurun_out/r2aj; export TMPDIR=/tmp
O=gpurun_out/r2aj
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2aj/bench_a.json").read().strip().splitlines()[-1])
print(round(d["value"], 1), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k != "note"})
PY
