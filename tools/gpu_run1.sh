set -x
cd $GRAFT_REPO_ROOT
python tools/probe_mfma_order.py > gpurun_out/probe.txt 2>&1
tail -12 gpurun_out/probe.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/pytest1.txt 2>&1
tail -40 gpurun_out/pytest1.txt
timeout 900 python bench.py > gpurun_out/bench1.txt 2>&1
tail -3 gpurun_out/bench1.txt
FALCON_BENCH_DEVICE=0 FALCON_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --no-configs > gpurun_out/bench2.txt 2>&1
tail -3 gpurun_out/bench2.txt
