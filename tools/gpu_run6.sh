cd $GRAFT_REPO_ROOT
timeout 900 python tools/fused_phases.py 1000000 0 > gpurun_out/phases.txt 2>&1
grep -v amdgpu.ids gpurun_out/phases.txt | tail -40
