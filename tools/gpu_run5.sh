cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fused.py -q --timeout 600 > gpurun_out/pytest5.txt 2>&1
tail -3 gpurun_out/pytest5.txt
timeout 900 python tools/fused_phases.py 1000000 0 > gpurun_out/phases.txt 2>&1
grep -E "dbg|staged|rows|members" gpurun_out/phases.txt
