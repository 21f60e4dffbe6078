#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python tools/nn_probe.py 1000000 10000000 '[{}]' > gpurun_out/probe.log 2>&1; cut -c1-900 gpurun_out/probe.log | tail -12
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
timeout 600 python bench.py --cold > gpurun_out/bench_cold.log 2>&1; tail -1 gpurun_out/bench_cold.log
