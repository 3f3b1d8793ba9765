cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ivf16.py -x -q --timeout 900 > gpurun_out/pytest15.txt 2>&1
tail -15 gpurun_out/pytest15.txt
timeout 900 python -m pytest tests/test_gpu_regimes.py -x -q --timeout 600 > gpurun_out/pytest15b.txt 2>&1
tail -5 gpurun_out/pytest15b.txt
timeout 600 python tools/scale_run.py 10000000 > gpurun_out/scale_ivf16.txt 2>&1
tail -4 gpurun_out/scale_ivf16.txt
bash tools/prof10m.sh 10000000 2>&1 | grep "ms total" | head -14
