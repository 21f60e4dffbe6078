#!/bin/bash
out=gpurun_out/r5w; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_two_ranks.py tests/test_gpu_comm.py tests/test_gpu_boundary_hostpath.py tests/test_gpu_scratch.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > $out/pytest_a.log 2>&1
echo "a rc=$? $(tail -1 $out/pytest_a.log)" | tee -a $out/rc.txt
timeout 600 python -m pytest tests -x -q -m perf -p no:cacheprovider > $out/pytest_perf.log 2>&1
echo "perf rc=$? $(tail -1 $out/pytest_perf.log)" | tee -a $out/rc.txt
tail -5 $out/pytest_a.log
for c in c2; do timeout 300 python bench.py --config $c --steps 40 --warmup 5 2>$out/$c.err | grep '^{"metric"' > $out/$c.json; python -c "
import json; d=json.loads(open('$out/$c.json').read()); print('$c', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d.get('kernel_ms'))"; done
