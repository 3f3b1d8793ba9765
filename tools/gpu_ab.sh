# quick A/B on the GPU box: parity tests touched by the change under test, then 10 M float32 timing (both bucket regimes)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest ${1:-tests/test_gpu_ivf16.py tests/test_gpu_regimes.py tests/test_gpu_fused.py} -x -q --timeout 900 > gpurun_out/pytest_ab.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/pytest_ab.txt | tail -6
timeout 240 python tools/scale_run.py 10000000 2>&1 | tail -2
timeout 240 python tools/scale_run.py 10000000 f32 32 f32 400 400 600 2>&1 | tail -2
