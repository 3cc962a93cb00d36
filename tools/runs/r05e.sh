# Round 5: the ping-pong loop as the default of eligible shapes — full GPU suite, then same-box A/B of the step against the round-4 kernels (diag: MVPTR_NT_EXP=131072)
O=gpurun_out/r05e; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for i in 1 2; do
MVPTR_LIB=diag python3 bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_nt8_$i.log 2>&1; tail -1 $O/bench_nt8_$i.log | cut -c1-200
MVPTR_LIB=diag MVPTR_NT_EXP=131072 python3 bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_old_$i.log 2>&1; tail -1 $O/bench_old_$i.log | cut -c1-200
done
MVPTR_LIB=diag python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length > $O/bench_fixed_nt8.log 2>&1; tail -1 $O/bench_fixed_nt8.log | cut -c1-200
MVPTR_LIB=diag MVPTR_NT_EXP=131072 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length > $O/bench_fixed_old.log 2>&1; tail -1 $O/bench_fixed_old.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --one-stream > $O/bench_one.log 2>&1; tail -1 $O/bench_one.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --model single > $O/bench_single.log 2>&1; tail -1 $O/bench_single.log | cut -c1-200
