O=gpurun_out/r02s; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -8 $O/gputest.log | cut -c1-400
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500
