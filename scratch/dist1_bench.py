# run bench.py under a 1-rank RCCL group with WORLD_SIZE faked to look like a multi-rank job is not possible;
# instead: RANK=0 WORLD_SIZE=1 with --force-exchange exercises everything but the collective itself.
