#!/bin/bash
# full GPU suite, then the round-2 evidence of scene A
mkdir -p gpurun_out/r2u; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r2u/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2u/pytest.log; tail -3 gpurun_out/r2u/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2u/smoke.log 2>&1; tail -1 gpurun_out/r2u/smoke.log
bash tools/gpu_r2_prof.sh
