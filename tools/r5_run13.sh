#!/bin/bash
out=gpurun_out/r5x; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1
echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -2 $out/pytest_pt2pt.log
timeout 1200 python tools/ab_probe.py $out/ab.json "default:" "norefine:refine_min=100000000" "refine64:refine_min=64" "refine32:refine_min=32" "refine256:refine_min=256" "refine1:refine_min=1" > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
timeout 400 python tools/timeline_probe.py > $out/timeline.json 2> $out/timeline.err
