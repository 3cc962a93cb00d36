O=gpurun_out/r02an; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|^FAILED" $O/gputest.log | tail -8
