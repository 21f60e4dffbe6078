#!/bin/bash
# round-end check of the committed build: full GPU suite + smoke
mkdir -p gpurun_out/r2final; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r2final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2final/pytest.log; tail -3 gpurun_out/r2final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2final/smoke.log 2>&1; tail -1 gpurun_out/r2final/smoke.log
