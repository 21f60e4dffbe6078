#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_matcher_pt2pl.py -x -q -m gpu -p no:cacheprovider -k "large_layer" --durations=3 > $O/pytest.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; tail -15 $O/pytest.log
