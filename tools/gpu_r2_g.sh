#!/bin/bash
# round 2, call G: the host path of the boundary (adapter host layer through its C ABI)
mkdir -p gpurun_out/r2g; export TMPDIR=/tmp
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_boundary_hostpath.py tests/test_abi.py -q -x --timeout=600 -s --durations=0 > $O/pytest_hostpath.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hostpath.log; tail -40 $O/pytest_hostpath.log
