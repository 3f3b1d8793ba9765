cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_ivf16.py tests/test_gpu_regimes.py -x -q --timeout 900 > gpurun_out/pytest26.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/pytest26.txt | tail -6
timeout 600 python tools/scale_run.py 10000000 2>&1 | tail -2
