#!/bin/bash
# tools/gpu_round_end.sh: the round's evidence in ONE bounded gpurun call (≈10 GPU-minutes): full -m gpu suite, smoke,
# the default bench line, the --config lines, the sharded C3 line on a one-rank communicator, the rocprofv3 summaries
# (kernel stats + HBM counters) of scene B / scene A / C3, the SQ counters of the point-to-plane kernels, and the
# timelines of the shipped search kernels.  Everything under gpurun_out/r3g/; copied to profiles/ afterwards by hand.
O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp
tools/gpu_run.sh r3g tests smoke bench > $O/run.txt 2>&1; cat $O/rc.txt
for c in c2 c3 c5; do timeout 300 python bench.py --config $c --steps 40 --warmup 5 2>/dev/null | grep '^{"metric"' > $O/$c.json; done
MP2P_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config c3 --steps 20 --warmup 3 2>$O/c3_dist.err | grep '^{"metric"' > $O/c3_dist.json
tools/gpu_prof.sh r03_bench_scene_b --scene b > /dev/null 2>&1
tools/gpu_prof.sh r03_bench_scene_a --scene a > /dev/null 2>&1
tools/gpu_prof.sh r03_bench_c3 --config c3 > /dev/null 2>&1
tools/gpu_pmc.sh r03_k5 "tools/pl_one.py 120000" pt2pl_ stats sq mem > /dev/null 2>&1
timeout 300 python tools/timeline_probe.py 1000000 10000000 b > $O/timeline_k3_scene_b.json 2>/dev/null
timeout 300 python tools/pl_timeline.py 120000 0.25 2>/dev/null | tail -1 > $O/timeline_k5_c3.json
du -sh gpurun_out; ls $O
