O=gpurun_out/r02ar; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "configs3" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|configs\[3\]\|assert" $O/gputest.log | tail -12
