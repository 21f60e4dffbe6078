#!/bin/bash
# tools/gpu_round_end.sh: the round's evidence in ONE bounded gpurun call: full -m gpu suite, smoke, the default bench line,
# the --config lines, the sharded C3 line on a one-rank communicator, the rocprofv3 summaries (kernel stats + HBM counters)
# of scene B / scene A / C3, the timeline of the shipped search kernels.  Everything under gpurun_out/r4g/; copied to
# profiles/ afterwards by tools/summarize_prof.py and by hand.
O=gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "tests rc=$? $(tail -1 $O/pytest_gpu.log)" | tee -a $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
for c in c2 c3 c5; do timeout 300 python bench.py --config $c --steps 40 --warmup 5 2>/dev/null | grep '^{"metric"' > $O/$c.json; done
MP2P_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config c3 --steps 20 --warmup 3 2>$O/c3_dist.err | grep '^{"metric"' > $O/c3_dist.json
tools/gpu_prof.sh r04_bench_scene_b --scene b > /dev/null 2>&1
tools/gpu_prof.sh r04_bench_scene_a --scene a > /dev/null 2>&1
tools/gpu_prof.sh r04_bench_c3 --config c3 > /dev/null 2>&1
timeout 300 python tools/timeline_probe.py 1000000 10000000 b > $O/timeline_k3_scene_b.json 2>/dev/null
du -sh gpurun_out; cat $O/rc.txt; tail -c 400 $O/bench.json
