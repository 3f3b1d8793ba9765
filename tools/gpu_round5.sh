#!/bin/bash
# The round's measurement run on the GPU box: bench (default command), kernel traces of the headline steps in both partition
# schedules (one stream / two streams), the 10 M float32 kernel trace.  Outputs under gpurun_out/ (copy to profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python bench.py > $O/r5_bench.txt 2>&1
grep '^{' $O/r5_bench.txt | tail -1 > $O/r5_bench_1M.json
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 bench.py --partitions pipelined --no-configs --no-cpu-baseline > $O/r5_bench_pipelined_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_bench_1M_pipelined_kernel_stats.csv
grep '^{' $O/r5_bench_pipelined_under_rocprof.txt | tail -1 > $O/r5_bench_1M_pipelined_under_rocprof.json
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 bench.py --no-configs --no-cpu-baseline > $O/r5_bench_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_bench_1M_kernel_stats.csv
grep '^{' $O/r5_bench_under_rocprof.txt | tail -1 > $O/r5_bench_1M_under_rocprof.json
rm -rf $O/p_trace
python3 - <<'PY'
import json
j = json.load(open('gpurun_out/r5_bench_1M.json'))
print('value', j['value'], j['ms_per_step'], j['stage_ms'], j.get('value_host_to_host'), j.get('ms_per_step_host_to_host_latency'))
print('roofline', {k: j['roofline'][k] for k in ('bound', 'frac', 'avg_launch_ms', 'traffic')})
for c in j.get('configs', []):
    print(c.get('dtype'), c.get('ms_per_step'), c.get('stage_ms'), c.get('error'), {k: c['roofline'][k] for k in ('bound', 'frac', 'avg_launch_ms')} if 'roofline' in c else None)
print(j.get('cpu_baseline'))
j = json.load(open('gpurun_out/r5_bench_1M_pipelined_under_rocprof.json')); print('pipelined under rocprof', j['value'], j['roofline']['avg_launch_ms'])
PY
grep "dense4_kernel<50>" $O/r5_bench_1M_pipelined_kernel_stats.csv | cut -c1-60,170-260
grep "dense4_kernel<50>" $O/r5_bench_1M_kernel_stats.csv | cut -c1-60,170-260
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 tools/scale_run.py 10000000 > $O/r5_10M_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_10M_f32_kernel_stats.csv
tail -2 $O/r5_10M_under_rocprof.txt
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 tools/scale_run.py 10000000 f32 16 f16 800 > $O/r5_10M_f16_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_10M_f16_800_kernel_stats.csv
tail -2 $O/r5_10M_f16_under_rocprof.txt
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 tools/scale_run.py 10000000 f32 32 f32 400 400 600 > $O/r5_10M_c4_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_10M_f32_dense_kernel_stats.csv
tail -2 $O/r5_10M_c4_under_rocprof.txt
rm -rf $O/p_trace
rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 tools/scale_run.py 10000000 f32 32 f32 400 400 560 65536 > $O/r5_10M_b64k_under_rocprof.txt 2>&1
python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_10M_f32_b64k_kernel_stats.csv
tail -2 $O/r5_10M_b64k_under_rocprof.txt
rm -rf $O/p_trace
