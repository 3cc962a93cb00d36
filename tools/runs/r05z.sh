# Round 5: multi-run loss curves of the 8-bit against the bf16 gelu' stash
O=gpurun_out/r05z; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/stash_soak.py --runs 4 --steps 3000 > $O/soak.log 2>&1; grep -v amdgpu $O/soak.log | cut -c1-400 | tail -14
python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm_tn_stack" 2>&1 | tail -2
