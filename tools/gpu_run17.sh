cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ivf16.py tests/test_gpu_regimes.py -x -q --timeout 900 > gpurun_out/pytest17.txt 2>&1
tail -5 gpurun_out/pytest17.txt
timeout 600 python tools/scale_run.py 10000000 2>&1 | tail -6
FALCON_PROBE_TABLE_UNORDERED=1 timeout 600 python tools/scale_run.py 10000000 2>&1 | tail -6
bash tools/prof10m.sh 10000000 2>&1 | grep "ms total" | head -8
