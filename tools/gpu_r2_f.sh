#!/bin/bash
# round 2, call F: BASELINE configs C4, C5 at full size (C2, C3 passed in the previous call)
mkdir -p gpurun_out/r2f; export TMPDIR=/tmp
O=gpurun_out/r2f
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --timeout=1200 --durations=0 -k "c4 or c5" > $O/pytest_configs2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_configs2.log; tail -40 $O/pytest_configs2.log
