#!/bin/bash
# tools/gpu_round_end.sh [tag]: the round's evidence in ONE bounded gpurun call: full -m gpu suite, smoke, the default bench
# line (parity gate included), the --config lines, the sharded C3 line on a one-rank communicator, the rocprofv3 summaries
# (kernel stats + HBM counters) of scene B / C3 / C2 / C5, the speed-of-light decomposition and the timeline of the shipped
# search kernels.  Everything under gpurun_out/<tag>/; copied to profiles/ afterwards by tools/summarize_prof.py and by hand.
T=${1:-r5end}
O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)" | tee -a $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
# the counter captures first, summarised on the box, so that the lines below carry the traffic of THIS build (bench.py refuses a
# capture whose kernel-source fingerprint differs); the summaries are redone from the merged gpurun_out/ afterwards
tools/gpu_prof.sh r05_bench_scene_b --scene b > /dev/null 2>&1
tools/gpu_prof.sh r05_bench_c3 --config c3 > /dev/null 2>&1
tools/gpu_prof.sh r05_bench_c2 --config c2 > /dev/null 2>&1
tools/gpu_prof.sh r05_bench_c5 --config c5 > /dev/null 2>&1
for t in scene_b c2 c3 c5; do python tools/summarize_prof.py r05_bench_$t r05_bench_$t > /dev/null 2>&1; done
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
for c in c2 c3 c5; do timeout 400 python bench.py --config $c --steps 40 --warmup 5 2>/dev/null | grep '^{"metric"' > $O/$c.json; done
MP2P_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config c3 --steps 20 --warmup 3 2>$O/c3_dist.err | grep '^{"metric"' > $O/c3_dist.json
timeout 600 python tools/ab_probe.py $O/sol.json "default:" --sol > $O/sol.txt 2> $O/sol.err
timeout 300 python tools/timeline_probe.py 1000000 10000000 b > $O/timeline_k3_scene_b.json 2>/dev/null
du -sh gpurun_out; cat $O/rc.txt; tail -c 300 $O/bench.json; cat $O/sol.txt
