timeout 900 python -m pytest tests/test_gpu_matcher_pt2pl.py tests/test_gpu_configs.py tests/test_gpu_matcher_adaptive.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
for t in "pl_warm=1" "pl_warm=0"; do
MP2P_HIP_TUNE=$t python bench.py --config c3 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t c3: it/s=%.0f ms=%.3f'%(d['value'],d['ms_per_step']), d['kernel_ms'], d['step_ms'], d['final_pose_error'])"
done
for t in "pl_warm=1" "pl_warm=0"; do
MP2P_HIP_TUNE=$t python bench.py --config c5 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t c5: it/s=%.0f ms=%.3f'%(d['value'],d['ms_per_step']), d['kernel_ms'], d['step_ms'])"
done
