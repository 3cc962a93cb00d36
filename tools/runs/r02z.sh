O=gpurun_out/r02z; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log | cut -c1-300
