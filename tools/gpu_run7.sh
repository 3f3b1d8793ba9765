cd $GRAFT_REPO_ROOT
timeout 600 python tools/fused_debug.py 128 > gpurun_out/debug.txt 2>&1
grep -v amdgpu.ids gpurun_out/debug.txt | tail -50
