cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_linkage.py -q --timeout 600 > gpurun_out/pytest9.txt 2>&1
tail -25 gpurun_out/pytest9.txt
