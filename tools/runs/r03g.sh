O=gpurun_out/r03g; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -v Gloo $O/gputest.log | tail -6 | cut -c1-400
grep -E "^PARITY" $O/gputest.log | sort | uniq > $O/parity_bounds.txt; cat $O/parity_bounds.txt
bash tools/runs/r03_profile.sh > $O/profile.log 2>&1
tail -120 $O/profile.log | cut -c1-250
