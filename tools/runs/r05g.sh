# Round 5: tile height by CU rounds for the stacks that own the GPU only (desc.beside) vs 256-row tiles everywhere, same box alternating
O=gpurun_out/r05g; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "gemm_nt or deferred or packed_training or stash or dropout" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export MVPTR_LIB=diag
for i in 1 2 3; do
for v in hint:0 mt8:33554432; do
  n=${v%%:*}; e=${v##*:}
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_${n}_$i.log 2>&1; echo -n "$n $i: "; tail -1 $O/bench_${n}_$i.log | cut -c150-200
done
done
for v in hint:0 mt8:33554432; do
  n=${v%%:*}; e=${v##*:}
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length > $O/bench_fixed_${n}.log 2>&1; echo -n "fixed $n: "; tail -1 $O/bench_fixed_${n}.log | cut -c150-200
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 --model single > $O/bench_single_${n}.log 2>&1; echo -n "single $n: "; tail -1 $O/bench_single_${n}.log | cut -c150-200
done
