#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --n-local 120000 --n-global 2000000 --steps 10 --warmup 3 --cpu-sample 120000 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err; echo "bench rc=$?" >> gpurun_out/bench_small.err
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_big.json 2> gpurun_out/bench_big.err; echo "bench rc=$?" >> gpurun_out/bench_big.err
tail -3 gpurun_out/smoke.log; tail -5 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/bench_small.err; cat gpurun_out/bench_small.json; tail -4 gpurun_out/bench_big.err; cat gpurun_out/bench_big.json
