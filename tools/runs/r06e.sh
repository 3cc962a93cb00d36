cd $GRAFT_REPO_ROOT
for v in "" bisA r05; do echo "== lib '$v'"; MVPTR_LIB=$v python3 tools/debug_gelu_pos.py 2>&1 | grep -v amdgpu | tail -12; done
