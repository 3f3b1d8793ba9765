#!/bin/bash
# Round 5 kernel traces (rocprofv3 --kernel-trace --stats) of the 10 M configurations -> gpurun_out/r5_*_kernel_stats.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
tr() {  # name, scale_run args...
  name=$1; shift
  rm -rf $O/p_trace
  rocprofv3 --kernel-trace --stats -d $O/p_trace -o t -- python3 tools/scale_run.py "$@" > $O/r5_${name}_under_rocprof.txt 2>&1
  python3 profiles/summarize.py stats $O/p_trace/t_results.db $O/r5_${name}_kernel_stats.csv
  tail -1 $O/r5_${name}_under_rocprof.txt | cut -c1-250
  rm -rf $O/p_trace
}
tr 10M_f32 10000000
tr 10M_f32_b64k 10000000 f32 32 f32 400 400 560 65536
[ "$1" = "all" ] && tr 10M_f16_800 10000000 f32 16 f16 800
[ "$1" = "all" ] && tr 10M_f32_dense 10000000 f32 32 f32 400 400 600
