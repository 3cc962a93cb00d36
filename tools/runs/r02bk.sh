cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bk
timeout 300 python tools/split_joint.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02bk/split.log
timeout 300 python tools/split_joint.py fixed 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02bk/split.log
