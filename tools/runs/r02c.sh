O=gpurun_out/r02c; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "rc=$?" >> $O/bench.log
grep -E "FAILED|passed|failed" $O/gputest.log | tail; tail -2 $O/bench.log | cut -c1-400
