bash tools/gpu_r3.sh r3r tests
echo "== wave_kernel=1 over the point-matcher tests"
MP2P_HIP_TUNE=wave_kernel=1 timeout 900 python -m pytest tests/test_gpu_matcher_pt2pt.py tests/test_gpu_fuzz.py tests/test_gpu_icp.py tests/test_gpu_fullsize_properties.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
