#!/bin/bash
# round 2, call J: communicator tests, GN weight-block test, pt2pl kernel probe on the C3 scene
mkdir -p gpurun_out/r2j; export TMPDIR=/tmp
O=gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_gn.py tests/test_gpu_boundary_hostpath.py -q -x --timeout=600 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
timeout 300 python tools/pl_probe.py 120000 10000000 0 b > $O/pl_probe.log 2>&1; echo "pl_probe rc=$?"; cat $O/pl_probe.log | cut -c1-400
