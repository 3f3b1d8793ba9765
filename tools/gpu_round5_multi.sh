#!/bin/bash
# Round 5, multi-GPU readiness on ONE GPU: every rank's share of the jobs in turn (tools/shard_share.py), and RCCL with two
# ranks on one device (expected to be refused: recorded either way).  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
O=gpurun_out
for w in 2 4 8; do timeout 600 python tools/shard_share.py $w strong > $O/r5_shard_share_${w}gpu_strong.txt 2>&1; tail -1 $O/r5_shard_share_${w}gpu_strong.txt | cut -c1-400; done
timeout 600 python tools/shard_share.py 8 weak > $O/r5_shard_share_8gpu_weak.txt 2>&1; tail -1 $O/r5_shard_share_8gpu_weak.txt | cut -c1-300
timeout 600 python tools/shard_share.py 8 skew > $O/r5_shard_share_8gpu_skew.txt 2>&1; tail -1 $O/r5_shard_share_8gpu_skew.txt | cut -c1-300
# two RCCL ranks on the one device
FALCON_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --exchange-only > $O/r5_rccl_two_ranks_one_gpu.txt 2>&1
echo "rccl 2 ranks on one GPU: rc=$?"; tail -5 $O/r5_rccl_two_ranks_one_gpu.txt | cut -c1-300
