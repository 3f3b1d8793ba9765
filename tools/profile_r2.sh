# Round-2 profiles of the headline command (run on the GPU box through gpurun; summaries are copied into profiles/).
#   kernel trace + stats of `python3 bench.py` (default flags), then FETCH_SIZE / WRITE_SIZE in separate PMC passes
#   (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass) of a shorter run, staged and with --prefilter.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/p_trace $O/p_f $O/p_w $O/p_ff $O/p_fw
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 bench.py --no-cpu-baseline > $O/r2_bench_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r2_bench_kernel_stats.csv
grep '^{' $O/r2_bench_under_rocprof.txt | tail -1 > $O/r2_bench_under_rocprof.json
CMD="python3 bench.py --steps 5 --warmup 1 --no-configs --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p_w -o w -- $CMD > /dev/null 2>&1
python3 profiles/summarize.py pmc $O/p_f/f_results.db $O/p_w/w_results.db 17 $O/r2_pmc_hbm_traffic_per_step.json "$CMD"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_ff -o f -- $CMD --prefilter > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p_fw -o w -- $CMD --prefilter > /dev/null 2>&1
python3 profiles/summarize.py pmc $O/p_ff/f_results.db $O/p_fw/w_results.db 17 $O/r2_pmc_hbm_traffic_per_step_prefilter.json "$CMD --prefilter"
rm -rf $O/p_trace $O/p_f $O/p_w $O/p_ff $O/p_fw
python3 - <<'PY'
import json
for f in ("r2_pmc_hbm_traffic_per_step.json", "r2_pmc_hbm_traffic_per_step_prefilter.json"):
    j = json.load(open("gpurun_out/" + f))
    print(f, {k: round(v["hbm_bytes"] / 1e9, 3) for k, v in j.items() if isinstance(v, dict)})
PY
head -12 $O/r2_bench_kernel_stats.csv | cut -c1-150
