#!/bin/bash
out=gpurun_out/r5s; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1
echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -2 $out/pytest_pt2pt.log
timeout 1200 python tools/ab_probe.py $out/ab.json "default:" "grp8:grp_all_bricks=8" "grp12:grp_all_bricks=12" "grp16:grp_all_bricks=16" "grp0:grp_all_bricks=0" > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
