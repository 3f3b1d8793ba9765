cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fused.py -x -q --timeout 600 > gpurun_out/pytest5.txt 2>&1
tail -15 gpurun_out/pytest5.txt
timeout 900 python tools/fused_phases.py 1000000 0,64,65,66,68,72,80,100,108,124,95 > gpurun_out/phases.txt 2>&1
grep -E "dbg|staged|rows" gpurun_out/phases.txt
