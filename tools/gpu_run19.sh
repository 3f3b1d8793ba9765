cd $GRAFT_REPO_ROOT
timeout 600 python tools/concurrent_parts.py 1000000 2>&1 | tail -3
timeout 600 python tools/concurrent_parts.py 2000000 2>&1 | tail -3
