cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_search.py -q --timeout 600 > gpurun_out/pytest13.txt 2>&1
tail -4 gpurun_out/pytest13.txt
for sr in 0 64 96; do
  FALCON_FINE_SHORT_ROWS=$sr timeout 600 python tools/scale_run.py 10000000 > gpurun_out/scale_$sr.txt 2>&1
  echo "short_rows=$sr"; tail -4 gpurun_out/scale_$sr.txt
done
