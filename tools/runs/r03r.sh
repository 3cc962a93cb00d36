cd $GRAFT_REPO_ROOT
python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "--- launch_bounds(256,3)"
MVPTR_LIB=occ3 python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
