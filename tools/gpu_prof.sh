#!/bin/bash
# tools/gpu_prof.sh <tag> [bench.py arguments]     (run on the GPU box through gpurun)
# The three rocprofv3 passes the judged summaries come from, of ONE bench command: kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE each in a pass of its own (counters never together with a runtime trace).  Raw output under
# gpurun_out/<tag>/; afterwards, in the repo:  python tools/summarize_prof.py <tag> <tag>   writes
# profiles/<tag>_kernel_stats.csv and profiles/<tag>_hbm_pmc.csv.
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp
( cd $R && python -c "import bench; print(bench.csrc_sha16())" > $O/csrc_sha16.txt )
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/bench_write.log 2>&1; echo "write rc=$?"
cd $R
find $O -name "*_kernel_trace.csv" -size +8M -delete   # the per-dispatch trace is not needed once the stats exist
tail -n 1 $O/bench_kt.log | cut -c 1-400
