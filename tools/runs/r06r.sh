cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu -k "attention or dropout" 2>&1 | grep -v amdgpu | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -6 | cut -c1-300
echo "== new"; python3 tools/bench_attn.py 2>&1 | grep -v amdgpu | tail -14
echo "== prev"; MVPTR_LIB=prev python3 tools/bench_attn.py 2>&1 | grep -v amdgpu | tail -14
for i in 1 2 3; do
  python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['ms_per_step'])"
  MVPTR_LIB=prev python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('prev', d['ms_per_step'])"
done
