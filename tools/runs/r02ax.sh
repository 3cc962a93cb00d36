O=gpurun_out/r02ax; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
MVPTR_TN_SLAB=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_slab.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_atomic.log
done
MVPTR_TN_SLAB=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_slab.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_atomic.log
P="--kernel-trace --stats --output-format csv"
MVPTR_TN_SLAB=1 rocprofv3 $P -d $O/mix_slab -o mix_slab -- python3 tools/prof_dominant.py 6 > $O/mix.log 2>&1
rocprofv3 $P -d $O/mix_atomic -o mix_atomic -- python3 tools/prof_dominant.py 6 >> $O/mix.log 2>&1
grep "gemm_tn\|tn_reduce" $O/mix_slab/mix_slab_kernel_stats.csv | cut -c1-160
grep "gemm_tn\|tn_reduce" $O/mix_atomic/mix_atomic_kernel_stats.csv | cut -c1-160
