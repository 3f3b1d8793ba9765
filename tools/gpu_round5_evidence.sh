#!/bin/bash
# Round 5 parity evidence on the final tree: the GPU suite under FALCON_DEBUG_POISON=1, randomised end-to-end parity
# (tools/fuzz_parity.py, incl. skewed datasets and 1,024-list buckets), the build / search stress at raised repetitions.
cd $GRAFT_REPO_ROOT
O=gpurun_out
(FALCON_DEBUG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $O/r5_gpu_suite_under_poison.txt; tail -2 $O/r5_gpu_suite_under_poison.txt
timeout 1500 python tools/fuzz_parity.py 70 50505 > $O/r5_fuzz_parity.txt 2>&1; tail -1 $O/r5_fuzz_parity.txt; grep -c "^OK" $O/r5_fuzz_parity.txt; grep "n_list_max=1024\|BAD" $O/r5_fuzz_parity.txt | cut -c1-160 | head
(FALCON_STRESS_REPS=600 timeout 1500 python -m pytest tests/test_gpu_stress.py -q 2>&1 | tail -2) > $O/r5_stress_600.txt; tail -1 $O/r5_stress_600.txt
