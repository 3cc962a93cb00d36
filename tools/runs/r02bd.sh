O=gpurun_out/r02bd; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|^FAILED" $O/gputest.log | tail -8
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_2.log
MVPTR_HEADS_BESIDE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_1.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_2.log
MVPTR_HEADS_BESIDE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_1.log
