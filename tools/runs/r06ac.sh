cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
export MVPTR_LIB=diag
for vg in 0 1 5 2 6; do
  echo "== MVPTR_ATTN_VG=$vg"
  MVPTR_ATTN_VG=$vg python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v amdgpu | grep -E "passed|failed|^E  " | tail -3 | cut -c1-200
  MVPTR_ATTN_VG=$vg python3 tools/bench_attn.py 2>&1 | grep -v amdgpu | grep -E "packed|B=512" | cut -c1-260
done
