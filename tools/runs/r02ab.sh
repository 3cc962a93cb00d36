O=gpurun_out/r02ab; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/fill_probe.py 256 2>&1 | grep -v amdgpu.ids | tee $O/fill256.log
timeout 600 python tools/fill_probe.py 512 2>&1 | grep -v amdgpu.ids | tee $O/fill512.log
