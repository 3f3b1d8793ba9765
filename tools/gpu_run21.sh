cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -o t -- python3 $R/bench.py --no-configs --no-cpu-baseline --steps 12 --warmup 2 > /tmp/o.txt 2>&1
grep '^{' /tmp/o.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
python3 $R/tools/timeline2.py /tmp/pt/t_results.db | tail -150
