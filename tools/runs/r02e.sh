O=gpurun_out/r02e; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
for v in prod exp exp2 exp3 exp4 exp5 prod; do timeout 300 python tools/exp_tn.py $v >> $O/ablate_tn.log 2>&1; done
grep -E "FAILED|passed|failed" $O/gputest.log | tail; cat $O/ablate_tn.log
