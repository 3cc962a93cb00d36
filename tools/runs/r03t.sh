O=gpurun_out/r03t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/gputest.log | tail -3
