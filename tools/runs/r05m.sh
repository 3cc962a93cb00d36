# Round 5: pipelined LayerNorm backward (two row sets in registers) against the plain loop, in the step and alone
O=gpurun_out/r05m; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or ln" > $O/pytest_ln.log 2>&1; tail -2 $O/pytest_ln.log
for k in 0 262144 256; do
  echo "== bench_ln MVPTR_NT_EXP=$k (0 pipelined 2 rows, 262144 plain loop, 256 pipelined 4 rows)"
  MVPTR_LIB=diag MVPTR_NT_EXP=$k python3 tools/bench_ln.py 2>/dev/null | grep "M="
done
for r in 1 2; do
for k in 0 262144 256; do
  echo "== bench packed, MVPTR_NT_EXP=$k" >> $O/ab.log
  MVPTR_LIB=diag MVPTR_NT_EXP=$k python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])" >> $O/ab.log 2>&1
done
done
cat $O/ab.log
