O=gpurun_out/r02bl; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "stream_placement or weight_cache" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|assert\|^E " $O/gputest.log | tail -12
