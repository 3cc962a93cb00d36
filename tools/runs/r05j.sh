# Round 5: blocked 8-bit stash + XCD-aligned remainder of the weight-gradient stack: correctness, then same-box A/B
O=gpurun_out/r05j; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for k in 0 2048; do
  echo "== tn stack, MVPTR_NT_EXP=$k (2048 = flat remainder of round 5a)" >> $O/tn.log
  MVPTR_LIB=diag MVPTR_NT_EXP=$k python3 tools/bench_tn_stack.py --reps 8 >> $O/tn.log 2>&1
done
cat $O/tn.log | grep -v "^knob"
for r in 1 2; do
for k in 0 2048 262144 264192; do
  echo "== bench packed, MVPTR_NT_EXP=$k (2048 flat remainder, 262144 row-major stash)" >> $O/ab.log
  MVPTR_LIB=diag MVPTR_NT_EXP=$k python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/ab.log 2>&1
done
done
cat $O/ab.log
python3 bench.py --steps 30 --warmup 8 --no-extras --fixed-length 2>/dev/null | tail -1 | cut -c1-200
