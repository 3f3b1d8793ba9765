cd $GRAFT_REPO_ROOT
timeout 900 python tools/fused_phases.py > gpurun_out/phases.txt 2>&1
cat gpurun_out/phases.txt | grep dbg
