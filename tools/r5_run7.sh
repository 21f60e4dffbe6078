#!/bin/bash
out=gpurun_out/r5q; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1
echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -4 $out/pytest_pt2pt.log
timeout 1200 python tools/ab_probe.py $out/ab.json "default:" "nosplit:split_cand=0" "split6000:split_cand=6000" "split8000:split_cand=8000" "split4000_t1024:split_cand=4000,split_tiles=1024" "split7000_grp12:split_cand=7000,grp_all_bricks=12" > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
timeout 400 python tools/timeline_probe.py > $out/timeline.json 2> $out/timeline.err
