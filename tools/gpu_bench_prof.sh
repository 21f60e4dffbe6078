#!/bin/bash
# full-size bench + rocprofv3 summaries (copied to profiles/ by hand afterwards)
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_big.json 2> gpurun_out/bench_big.err; echo "bench rc=$?" >> gpurun_out/bench_big.err; tail -3 gpurun_out/bench_big.err
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/bench_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/bench_fetch -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/bench_write -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof/bench_write.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -20
