#!/bin/bash
# round 2, call J2: communicator tests; pt2pl search/fit split (Q = 8 / 32): parity + probe + C3 line
mkdir -p gpurun_out/r2j; export TMPDIR=/tmp
O=gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_matcher_pt2pl.py tests/test_gpu_fuzz.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_icp.py -q -x --timeout=600 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x --timeout=600 -k "c3 or c5" > $O/pytest_cfg.log 2>&1; echo "pytest cfg rc=$?" >> $O/pytest_cfg.log; tail -4 $O/pytest_cfg.log
for q in 0 8 32; do MP2P_HIP_TUNE="pl_q=$q" timeout 300 python tools/pl_probe.py 120000 10000000 0 b > $O/pl_probe_q$q.log 2>&1; echo "pl_probe q=$q rc=$?"; grep -v knn_dbg $O/pl_probe_q$q.log | head -2 | cut -c1-300; grep knn_dbg $O/pl_probe_q$q.log | head -2 | cut -c1-300; done
for q in 8 32; do MP2P_HIP_TUNE="pl_q=$q" timeout 300 python bench.py --config c5 > $O/bench_c5_q$q.json 2> $O/bench_c5_q$q.err; python -c "
import json; e=json.loads(open('$O/bench_c5_q$q.json').read().strip().splitlines()[-1]); print('c5 q=$q', round(e['value'],1), round(e['ms_per_step'],3), e['kernel_ms'])"; done
