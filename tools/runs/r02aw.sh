cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aw
timeout 300 python tools/timeline_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02aw/tl.log
timeout 300 python tools/timeline_attn.py sorted 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02aw/tl_sorted.log
