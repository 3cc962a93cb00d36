O=gpurun_out/r03q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -8
python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "--- two-pass kernel (diag build)"
MVPTR_LIB=diag MVPTR_ATTN_TWO_PASS=1 python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
