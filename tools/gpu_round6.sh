#!/bin/bash
# Round 6 measurements on the GPU box -> gpurun_out/r6_* (copied to profiles/ by the builder).
#   bash tools/gpu_round6.sh bench    the default bench line (+ cold pass, configs)
#   bash tools/gpu_round6.sh traces   rocprofv3 --kernel-trace --stats of the headline (one stream) and the 10 M configurations
#   bash tools/gpu_round6.sh pmc      HBM traffic per launch of the cosine kernels (FETCH_SIZE / WRITE_SIZE, separate passes)
#   bash tools/gpu_round6.sh parity   GPU suite (plain + under FALCON_DEBUG_POISON=1), fuzz, stress
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
what=${1:-bench}
if [ "$what" = bench ]; then
  python bench.py > $O/r6_bench_1M.json 2> $O/r6_bench_1M.err; tail -2 $O/r6_bench_1M.err; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_1M.json").read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"], 3), "ms; h2h", round(d.get("ms_per_step_host_to_host", 0), 2), "cold", d.get("cold_ms"))
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "issued_frac", "avg_launch_ms")})
for e in d.get("configs", []):
    print(e.get("baseline_config", "")[:50], "|", round(e.get("ms_per_step", 0), 1), "ms", e.get("error", ""), e.get("shares_note", ""), e.get("stage_ms"))
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
fi
tr() {  # name, scale_run args...
  name=$1; shift
  rm -rf /tmp/p_trace
  rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o t -- python3 tools/scale_run.py "$@" > $O/r6_${name}_under_rocprof.txt 2>&1
  python3 profiles/summarize.py stats /tmp/p_trace/t_results.db $O/r6_${name}_kernel_stats.csv
  tail -1 $O/r6_${name}_under_rocprof.txt | cut -c1-250
  rm -rf /tmp/p_trace
}
if [ "$what" = traces ]; then
  rm -rf /tmp/p_trace
  rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o t -- python3 bench.py --partitions pipelined --no-configs --no-cpu-baseline --no-cold > $O/r6_bench_1M_pipelined_under_rocprof.json 2> /dev/null
  python3 profiles/summarize.py stats /tmp/p_trace/t_results.db $O/r6_bench_1M_pipelined_kernel_stats.csv
  head -4 $O/r6_bench_1M_pipelined_kernel_stats.csv | cut -c1-160
  tr 10M_f32 10000000
  tr 10M_f16_800 10000000 f32 16 f16 800
  tr 10M_f32_dense 10000000 f32 32 f32 400 400 600
  tr 10M_f32_b64k 10000000 f32 32 f32 400 400 560 65536
  tr 10M_f32_d200 10000000 f32 16 f32 200
fi
if [ "$what" = pmc ]; then
  FALCON_PMC_OUT=r6_pmc_traffic.json bash tools/pmc_traffic.sh
fi
if [ "$what" = parity ]; then
  (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3) > $O/r6_gpu_suite_final.txt; tail -1 $O/r6_gpu_suite_final.txt
  (FALCON_DEBUG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3) > $O/r6_gpu_suite_under_poison.txt; tail -1 $O/r6_gpu_suite_under_poison.txt
  timeout 1500 python tools/fuzz_parity.py 70 60606 > $O/r6_fuzz_parity.txt 2>&1; tail -1 $O/r6_fuzz_parity.txt; grep -c "^OK" $O/r6_fuzz_parity.txt
  (FALCON_STRESS_REPS=600 timeout 1500 python -m pytest tests/test_gpu_stress.py -q 2>&1 | tail -2) > $O/r6_stress_600.txt; tail -1 $O/r6_stress_600.txt
fi
