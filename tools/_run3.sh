cd $GRAFT_REPO_ROOT && timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Ceres Solver Report" | tail -40
