#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_p20; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/c5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/c3.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete
for c in c5 c3; do echo "== $c"; f=$(find $O/kt_$c -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-140; done
