#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
FALCON_STRESS_REPS=10 timeout 1200 python -m pytest tests/test_gpu_ivf16.py tests/test_gpu_regimes.py tests/test_gpu_stress.py tests/test_gpu_search.py -x -q 2>&1 | tail -3
timeout 600 python tools/scale_run.py 10000000 2>&1 | tail -4
