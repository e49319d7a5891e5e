cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/s9_pytest.log 2>&1; echo "pytest rc=$?"; tail -20 gpurun_out/s9_pytest.log
