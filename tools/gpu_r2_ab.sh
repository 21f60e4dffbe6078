#!/bin/bash
mkdir -p gpurun_out/r2ab; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_properties.py -q -x --timeout=600 > gpurun_out/r2ab/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2ab/pytest.log; tail -2 gpurun_out/r2ab/pytest.log
bash tools/gpu_r2_prof.sh
