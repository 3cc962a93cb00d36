O=gpurun_out/r02x; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_tn" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -12 $O/gputest.log | cut -c1-300
echo "== slab"; timeout 600 python tools/sweep_tn_group.py q 0 64000,37748,19200,10917 0 2>&1 | grep "^M=" | tee $O/sweep_slab.log
echo "== atomics"; MVPTR_TN_SLAB=0 timeout 600 python tools/sweep_tn_group.py q 0 64000,37748,19200,10917 0 2>&1 | grep "^M=" | tee $O/sweep_atomic.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_packed.log
MVPTR_TN_SLAB=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_packed_atomic.log
done
