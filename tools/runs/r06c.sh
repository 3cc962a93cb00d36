# Round 6, call 3: bisect the retrieval-train gradient difference (new python + old / new kernels), re-check the two other failures
O=gpurun_out/r06c; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== new kernels"; python3 tools/debug_ft_grads.py 2>&1 | grep -v amdgpu | tail -40
echo "== r05 kernels"; MVPTR_LIB=r05 python3 tools/debug_ft_grads.py 2>&1 | grep -v amdgpu | tail -40
python3 -m pytest tests -q -m gpu -k "gelu_epilogue_function or b64_vs_oracle or dropout_kernels_follow or compact_scored or sync_free_joint or non_f32" 2>&1 | grep -E "^FAILED|passed|failed|^E  " | tail -12
