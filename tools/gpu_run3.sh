set -x
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fused.py -x -q --timeout 600 > gpurun_out/pytest3.txt 2>&1
tail -30 gpurun_out/pytest3.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 30 > gpurun_out/bench3.txt 2>&1
tail -2 gpurun_out/bench3.txt | cut -c1-2500
