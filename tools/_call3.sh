#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_icp.py tests/test_gpu_gn.py tests/test_gpu_comm.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_multilayer.py tests/test_gpu_boundary_hostpath.py tests/test_gpu_horn.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)" | tee -a $O/rc.txt
tail -3 $O/pytest.log
timeout 1200 python tools/ab_probe.py $O/ab.json "f0:gn_fuse_first=0" "f1:gn_fuse_first=1" "f0b:gn_fuse_first=0" "f1b:gn_fuse_first=1" > $O/ab.txt 2> $O/ab.err; tail -6 $O/ab.txt | cut -c1-200
grep -h "final\|err=" $O/ab.err | tail -4 | cut -c1-200
