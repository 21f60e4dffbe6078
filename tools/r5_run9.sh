#!/bin/bash
out=gpurun_out/r5t; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multilayer.py -x -q -m gpu -p no:cacheprovider > $out/pytest_ml.log 2>&1
echo "multilayer rc=$? $(tail -1 $out/pytest_ml.log)" | tee -a $out/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize_properties.py -x -q -m gpu -p no:cacheprovider > $out/pytest_full.log 2>&1
echo "fullsize rc=$? $(tail -1 $out/pytest_full.log)" | tee -a $out/rc.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/rc.txt
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","parity_gate","value_through_host_containers","speedup_vs_cpu_baseline")})
print(d["kernel_ms"]); print(d["roofline"]); cb=d["cpu_baseline"]; print({k:cb[k] for k in cb if k!="single_thread"}); print(cb["single_thread"])
print(d.get("host_boundary"))
PY
