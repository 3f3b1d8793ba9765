"""Run bench.py's world_size=2 code path on ONE GPU with loop-back collectives (test rig only)."""
import os, sys, runpy
import torch, torch.distributed as dist
os.environ.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
dist.init_process_group = lambda *a, **k: None
dist.destroy_process_group = lambda *a, **k: None
dist.barrier = lambda *a, **k: None
dist.is_initialized = lambda: True
dist.get_rank = lambda *a, **k: 0
dist.get_world_size = lambda *a, **k: 2
def _all_gather(out, t, *a, **k):
    for o in out: o.copy_(t)
dist.all_gather = _all_gather
dist.all_reduce = lambda t, *a, **k: None
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--spectra", "200000"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
