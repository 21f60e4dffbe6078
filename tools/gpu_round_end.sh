#!/bin/bash
# tools/gpu_round_end.sh [tag]: the round's evidence in ONE bounded gpurun call: full -m gpu suite, smoke, the rocprofv3 summaries
# (kernel stats + HBM counters) of scene B / C3 / C2 / C5 FIRST (summarised on the box, so that the bench lines below carry the
# traffic of THIS build: bench.py refuses a capture whose kernel-source fingerprint differs), the default bench line (parity gate
# included), the --config lines, the sharded C3 line on a one-rank communicator, the timeline of the shipped search kernels and the
# point-to-plane pose-sequence fuzz campaign with the ball-rule kernel forced.  Everything under gpurun_out/<tag>/ (and the
# summaries under profiles/, merged back through gpurun_out/<tag>/profiles_out/).
T=${1:-r6end}; R=r06
O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)" | tee -a $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
tools/gpu_prof.sh ${R}_bench_scene_b --scene b > /dev/null 2>&1
tools/gpu_prof.sh ${R}_bench_c3 --config c3 > /dev/null 2>&1
tools/gpu_prof.sh ${R}_bench_c2 --config c2 > /dev/null 2>&1
tools/gpu_prof.sh ${R}_bench_c5 --config c5 > /dev/null 2>&1
for t in scene_b c2 c3 c5; do python tools/summarize_prof.py ${R}_bench_$t ${R}_bench_$t > /dev/null 2>&1; done
mkdir -p $O/profiles_out; cp profiles/${R}_bench_*_kernel_stats.csv profiles/${R}_bench_*_hbm_pmc.csv profiles/${R}_bench_*_hbm_pmc.meta.json $O/profiles_out/ 2>/dev/null
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
for c in c2 c3 c5; do timeout 500 python bench.py --config $c --steps 40 --warmup 5 2>/dev/null | grep '^{"metric"' > $O/$c.json; echo "$c rc=$?" | tee -a $O/rc.txt; done
MP2P_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config c3 --steps 20 --warmup 3 2>$O/c3_dist.err | grep '^{"metric"' > $O/c3_dist.json
timeout 300 python tools/timeline_probe.py 1000000 10000000 b > $O/timeline_k3_scene_b.json 2>/dev/null
MP2P_HIP_TUNE=pl_select=1 MP2P_FUZZ_PLSEQ_SEEDS=0:1500 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -k "pose_seq or plseq or sequence" > $O/fuzz_plseq_ball.log 2>&1; echo "fuzz plseq (ball kernel) rc=$? $(grep -E 'passed|failed' $O/fuzz_plseq_ball.log | tail -1)" | tee -a $O/rc.txt
find gpurun_out -name "*.csv" -size +6M -delete; find gpurun_out -name "*.db" -delete
du -sh gpurun_out; cat $O/rc.txt; tail -c 300 $O/bench.json
