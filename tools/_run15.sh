cd $GRAFT_REPO_ROOT && timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Ceres Solver Report" | tail -8 > gpurun_out/pytest_r04_c.log
bash tools/collect_evidence.sh r04_c > gpurun_out/evidence_r04_c.log 2>&1
bash tools/collect_cfg5.sh r04_c > gpurun_out/evidence5_r04_c.log 2>&1
tail -3 gpurun_out/pytest_r04_c.log
