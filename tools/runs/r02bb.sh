O=gpurun_out/r02bb; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
MVPTR_NT_EXP=32768 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_new.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_old.log
done
MVPTR_NT_EXP=32768 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_new.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_old.log
MVPTR_NT_EXP=32768 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_new.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | cut -c1-330 | tee -a $O/bench_fixed_old.log
