cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02as
timeout 300 python tools/gap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02as/gap.log
