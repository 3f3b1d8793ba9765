cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest_final.txt 2>&1
tail -6 $O/pytest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_final.txt 2>&1
grep '^{' $O/bench_final.txt | tail -1 > $O/r2_bench_1M.json
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 bench.py --no-cpu-baseline --no-configs > $O/r2_bench_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r2_bench_1M_kernel_stats.csv
grep '^{' $O/r2_bench_under_rocprof.txt | tail -1 > $O/r2_bench_1M_under_rocprof.json
rm -rf $O/p_trace
FALCON_BENCH_DEVICE=0 FALCON_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_2ranks_gloo.txt 2>&1
grep '^{' $O/bench_2ranks_gloo.txt | tail -1 > $O/r2_bench_2ranks_gloo_one_gpu.json
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r2_bench_1M.json')); print('value',j['value'],j['ms_per_step'],j['stage_ms'], j['value_host_to_host'])
for c in j['configs']: print(c['dtype'], c['ms_per_step'], c['stage_ms'])
print(j['cpu_baseline'])
j=json.load(open('gpurun_out/r2_bench_1M_under_rocprof.json')); print('rocprof run', j['value'], j['roofline']['avg_launch_ms'], j['roofline']['traffic'])
j=json.load(open('gpurun_out/r2_bench_2ranks_gloo_one_gpu.json')); print('2 ranks', j['value'], j['ms_per_step'])
PY
grep "dense_kernel<50, 0>" $O/r2_bench_1M_kernel_stats.csv | cut -c1-60,170-260
# 10 M float32 (the IVF regime): kernel trace + HBM traffic of three passes
bash tools/prof10m.sh 10000000 > $O/prof10m.txt 2>&1; cp /tmp/k.csv $O/r2_10M_f32_kernel_stats.csv; grep "ms total" $O/prof10m.txt | head -16
bash tools/pmc10m.sh 10000000 > $O/r2_10M_f32_pmc_hbm_traffic.txt 2>&1; tail -20 $O/r2_10M_f32_pmc_hbm_traffic.txt
