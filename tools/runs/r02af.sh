O=gpurun_out/r02af; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=" $O/gputest.log | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_packed.log
MVPTR_GEMM_TN=q timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_packed_q.log
done
bash tools/runs/r02_profile.sh > $O/profile.log 2>&1
tail -40 $O/profile.log | cut -c1-250
