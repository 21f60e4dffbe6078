#!/bin/bash
# tools/gpu_run.sh <tag> <steps...>   steps: smoke tests pt2pt bench bencha benchb   (through gpurun)
# one bounded GPU call made of named steps; everything lands under gpurun_out/<tag>/
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/rc.txt;;
    pt2pt)   timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1; echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -5 $out/pytest_pt2pt.log;;
    tests)   timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "tests rc=$?" | tee -a $out/rc.txt; tail -5 $out/pytest_gpu.log;;
    bench)   timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/rc.txt; tail -c 600 $out/bench.json;;
    bencha)  timeout 600 python bench.py --steps 20 --warmup 5 --scene a --no-extras --no-cpu-baseline > $out/bench_a.json 2> $out/bench_a.err; echo "bencha rc=$?" | tee -a $out/rc.txt;;
    benchb)  timeout 600 python bench.py --steps 20 --warmup 5 --scene b --no-extras --no-cpu-baseline > $out/bench_b.json 2> $out/bench_b.err; echo "benchb rc=$?" | tee -a $out/rc.txt;;
    *) echo "unknown step $step";;
  esac
done
grep -h "chain step [089]" $out/*.err 2>/dev/null | tail -20
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get("kernel_ms",{})
        print(f, "it/s=%.0f ms=%.3f nn=%.3f"%(d["value"],d["ms_per_step"],k.get("nn_search",0)), {a:round(b,3) for a,b in k.items() if isinstance(b,float)})
        for sc in ("scene_a","scene_b"):
            if sc in d and "value" in d[sc]: print("   ",sc,"it/s=%.0f"%d[sc]["value"], {a:round(b,3) for a,b in d[sc]["kernel_ms"].items()})
    except Exception as e: print(f,"ERR",e)
PY
