set -x
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 > gpurun_out/pytest2.txt 2>&1
tail -40 gpurun_out/pytest2.txt
