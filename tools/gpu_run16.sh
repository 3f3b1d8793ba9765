cd $GRAFT_REPO_ROOT
echo default; timeout 300 python tools/sort_ab.py 2>&1 | grep "n="
echo onesweep; FALCON_SORT_ONESWEEP=1 timeout 300 python tools/sort_ab.py 2>&1 | grep "n="
