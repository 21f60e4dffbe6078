#!/bin/bash
O=gpurun_out/r6_h7; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_matcher_pt2pt.py tests/test_gpu_icp.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_bench_two_ranks.py -x -q -m gpu -p no:cacheprovider -k "not gauss" > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -5 $O/pytest.log
timeout 1200 python tools/ab_probe.py $O/ab.json "base:hint_r_pct=0" "h200:hint_r_pct=200" "h300:hint_r_pct=300" "h400:hint_r_pct=400" "h600:hint_r_pct=600" "h1000:hint_r_pct=1000" "base2:hint_r_pct=0" "hard1000:hard_cand=1000" "hard2500:hard_cand=2500" "hard1700:hard_cand=1700" "sb4:single_blocks_per_cu=4" "sb10:single_blocks_per_cu=10" "sb40:single_blocks_per_cu=40" "grp4:grp_all_bricks=4" "grp8:grp_all_bricks=8" "grp6:grp_all_bricks=6" > $O/ab.txt 2> $O/ab.err; tail -20 $O/ab.txt | cut -c1-170
