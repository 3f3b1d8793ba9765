cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof10m
rocprofv3 --kernel-trace --stats -d gpurun_out/prof10m -o p -- python3 tools/prof10m.py 10000000 f32 > gpurun_out/prof10m.txt 2>&1
tail -2 gpurun_out/prof10m.txt
python3 profiles/summarize.py stats gpurun_out/prof10m/p_results.db gpurun_out/r2_10M_f32_kernel_stats.csv
rm -f gpurun_out/prof10m/p_results.db
head -30 gpurun_out/r2_10M_f32_kernel_stats.csv | cut -c1-160
