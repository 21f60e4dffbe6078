#!/bin/bash
mkdir -p gpurun_out/r2t; export TMPDIR=/tmp
O=gpurun_out/r2t
timeout 900 python -m pytest tests/test_gpu_boundary_hostpath.py -q -x -s --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -a "host path\]\|passed\|failed\|rc=" $O/pytest.log | cut -c1-600 | tail -8
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2t/bench_full.json").read().strip().splitlines()[-1])
print(round(d["value"], 1), d["ms_per_step"], json.dumps(d.get("host_boundary"))[:1500])
PY
