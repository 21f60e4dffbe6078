O=gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_abort_regression.py tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $O/pytest.log)" | tee -a $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --scene b --no-extras --no-cpu-baseline > $O/bench_b.json 2> $O/bench_b.err; echo "benchb rc=$?" | tee -a $O/rc.txt
MP2P_HIP_TUNE=tile_bricks=0,hard_cand=0 timeout 300 python bench.py --steps 20 --warmup 5 --scene b --no-extras --no-cpu-baseline > $O/bench_b_old.json 2> $O/bench_b_old.err; echo "benchb_old rc=$?" | tee -a $O/rc.txt
MP2P_HIP_TUNE=hard_cand=0 timeout 300 python bench.py --steps 20 --warmup 5 --scene b --no-extras --no-cpu-baseline > $O/bench_b_nocost.json 2> $O/bench_b_nocost.err; echo "benchb_nocost rc=$?" | tee -a $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --scene a --no-extras --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err; echo "bencha rc=$?" | tee -a $O/rc.txt
grep -h "chain step [05-9]" $O/bench_b.err | tail -6
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
        print(f, "it/s=%.0f ms=%.3f"%(d["value"],d["ms_per_step"]), {a:round(b,3) for a,b in k.items() if isinstance(b,float)}, {a:(round(b,3) if isinstance(b,float) else b) for a,b in d.get("nn_stats",{}).items()})
    except Exception as e: print(f,"ERR",e)
PY
