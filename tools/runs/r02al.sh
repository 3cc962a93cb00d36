cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02al
timeout 300 python tools/small_mm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02al/small_mm.log
