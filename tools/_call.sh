O=gpurun_out/r4n; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("it/s=%.0f ms=%.3f"%(d["value"],d["ms_per_step"]), json.dumps(d["roofline"])[:700])
PY
