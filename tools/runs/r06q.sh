cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
python3 -m pytest tests -x -q -m gpu -k "dropout or layernorm or gemm_nt or encoder_stack_with_dropout" 2>&1 | grep -v amdgpu | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -6 | cut -c1-300
python3 tools/blas_table.py --ms 37748 2>&1 | grep -v amdgpu | grep -E "RESID|sum" 
MVPTR_LIB=prev python3 tools/blas_table.py --ms 37748 2>&1 | grep -v amdgpu | grep -E "RESID|sum"
python3 tools/bench_ln.py 2>&1 | grep -v amdgpu | tail -12
MVPTR_LIB=prev python3 tools/bench_ln.py 2>&1 | grep -v amdgpu | tail -12
for i in 1 2 3; do
  python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['ms_per_step'])"
  MVPTR_LIB=prev python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('prev', d['ms_per_step'])"
done
