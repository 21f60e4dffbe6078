#!/bin/bash
# round 2: the evidence kept under profiles/ -- the default bench line as emitted, rocprofv3 kernel
# stats of the same command, and the HBM traffic counters (separate --pmc passes, no trace domains)
mkdir -p gpurun_out/r2k/prof; export TMPDIR=/tmp
O=gpurun_out/r2k
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof/bench_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof/bench_kt.log 2>&1; echo "rocprof kt rc=$?"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/prof/bench_fetch -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof/bench_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/prof/bench_write -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$O/prof/bench_write.log 2>&1; echo "rocprof write rc=$?"
cd $R
find $O/prof -name "*.csv" | head; tail -2 $O/bench_n1.err | cut -c1-200
