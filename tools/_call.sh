O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -k round4 > $O/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $O/pytest.log)" | tee -a $O/rc.txt
grep -v "^  File" $O/pytest.log | grep "Error\|assert\|FAILED\|info\|passed\|failed" | cut -c1-400 | tail -40
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("it/s=%.0f ms=%.3f"%(d["value"],d["ms_per_step"])); hb=d["host_boundary"]; print({k:hb[k] for k in ("ms_per_step","vs_device_resident_step")}, hb["stage_ms"]); print(json.dumps(d["roofline"])[:500])
PY
