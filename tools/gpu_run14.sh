cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ivf16.py -x -q --timeout 900 > gpurun_out/pytest14.txt 2>&1
tail -15 gpurun_out/pytest14.txt
timeout 900 python -m pytest tests/test_gpu_regimes.py -x -q --timeout 600 > gpurun_out/pytest14b.txt 2>&1
tail -5 gpurun_out/pytest14b.txt
timeout 600 python tools/scale_run.py 10000000 > gpurun_out/scale_ivf16.txt 2>&1
tail -4 gpurun_out/scale_ivf16.txt
FALCON_NO_IVF16=1 timeout 600 python tools/scale_run.py 10000000 > gpurun_out/scale_noivf16.txt 2>&1
tail -4 gpurun_out/scale_noivf16.txt
