#!/bin/bash
out=gpurun_out/r5e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1
echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -4 $out/pytest_pt2pt.log
timeout 1200 python tools/ab_probe.py $out/ab.json "default:" --sol > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
