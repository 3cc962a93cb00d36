# Round 5, final code: soak (changing variable-length batches; finite losses, flat memory)
O=gpurun_out/r05ac; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/soak.py > $O/soak.log 2>&1; grep -v amdgpu $O/soak.log | tail -6 | cut -c1-250
