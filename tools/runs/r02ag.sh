O=gpurun_out/r02ag; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=" $O/gputest.log | tail -3
bash tools/runs/r02_profile.sh > $O/profile.log 2>&1
tail -5 $O/profile.log | cut -c1-250
timeout 600 python bench.py 2>$O/bench_default.err | tail -1 | tee $O/bench_default.json | cut -c1-600
timeout 300 python bench.py --fixed-length --no-extras 2>/dev/null | tail -1 | tee $O/bench_fixed.json | cut -c1-300
