cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SFMBA_BUILD_TIMING=1 python $R/tools/time_shim_incremental.py 2>&1 | grep -v "Ceres Solver" | tail -34
