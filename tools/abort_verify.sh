#!/bin/bash
# tools/abort_verify.sh <tag> <loops>: the sequence that aborted in GPUTEST_r03 (golden + host-path + comm tests in one
# process), <loops> times with output capture off, then the regression tests of tests/test_gpu_abort_regression.py
tag=${1:-verify}; N=${2:-12}; O=gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp
FILES="tests/test_golden.py tests/test_gpu_boundary_hostpath.py tests/test_gpu_comm.py"
for i in $(seq 1 $N); do
  PYTHONFAULTHANDLER=1 timeout 300 python -m pytest $FILES -x -q -m gpu -s -p no:cacheprovider > /tmp/run.log 2>&1
  rc=$?; echo "loop $i rc=$rc $(tail -1 /tmp/run.log)" | tee -a $O/rc.txt
  [ $rc -ne 0 ] && tail -c 5000 /tmp/run.log > $O/fail_$i.tail
done
timeout 600 python -m pytest tests/test_gpu_abort_regression.py -x -q -m gpu -p no:cacheprovider > $O/regression.log 2>&1
echo "regression rc=$? $(tail -1 $O/regression.log)" | tee -a $O/rc.txt
