O=gpurun_out/r02v; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -8 $O/gputest.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-400 | tee -a $O/bench_packed.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-400 | tee $O/bench_fixed.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --model single 2>&1 | tail -1 | cut -c1-400 | tee $O/bench_single.log
timeout 300 python tools/cpu_enqueue.py 2>&1 | tail -30 | tee $O/cpu_enqueue.log
timeout 300 python tools/gpu_idle.py 2>&1 | tail -20 | tee $O/gpu_idle.log
