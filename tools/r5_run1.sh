#!/bin/bash
# round 5, GPU call 1: parity of the new search path, then A/B on the headline workload
out=gpurun_out/r5a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py -x -q -m gpu > $out/pytest_pt2pt.log 2>&1
echo "pt2pt rc=$?" | tee -a $out/rc.txt; tail -8 $out/pytest_pt2pt.log
timeout 900 python tools/ab_probe.py $out/ab.json "r4:tile_select=0" "sel_lane:tile_select=1,nn_direct=0" "sel_direct:tile_select=1,nn_direct=1" "sel_direct_cap12k:tile_select=1,nn_direct=1,tile_cand_cap=12288" --sol --tpc 0,3 > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
grep "\[ab\] map\|SOL" $out/ab.err | tail -8
