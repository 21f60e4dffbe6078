#!/bin/bash
mkdir -p gpurun_out/r2l; export TMPDIR=/tmp
O=gpurun_out/r2l
timeout 300 python tools/pl_q_debug.py > $O/pl_q_debug.log 2>&1; tail -8 $O/pl_q_debug.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_matcher_pt2pl.py tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_icp.py -q -x --timeout=600 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x --timeout=600 -k "c3 or c5" > $O/pytest_cfg.log 2>&1; echo "pytest cfg rc=$?" >> $O/pytest_cfg.log; tail -4 $O/pytest_cfg.log
for c in c3 c5; do timeout 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; e=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(e['value'],1), round(e['ms_per_step'],3), e['kernel_ms'])"; done
