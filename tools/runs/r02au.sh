cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02au
timeout 300 python tools/grad_repro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02au/dbg.log
