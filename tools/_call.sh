O=gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --scene b --no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/rc.txt; }
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $O/pytest.log)" | tee -a $O/rc.txt
EXTRA="" run coop4 X=1
EXTRA="" run coop0 MP2P_HIP_TUNE=coop_max=0
EXTRA="" run coop1 MP2P_HIP_TUNE=coop_max=1
EXTRA="--scene a" run coop0_a MP2P_HIP_TUNE=coop_max=0
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
        print(f, "it/s=%.0f ms=%.3f"%(d["value"],d["ms_per_step"]), {a:round(b,3) for a,b in k.items() if isinstance(b,float)}, "deferred=%.3f"%d["nn_stats"]["deferred_to_one_query_kernel_frac"])
    except Exception as e: print(f,"ERR",str(e)[:100])
PY
grep -v "^  File" $O/pytest.log | tail -8 | cut -c1-250
