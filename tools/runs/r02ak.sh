O=gpurun_out/r02ak; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/glue_ops.py 2>&1 | grep -v amdgpu.ids | tee $O/glue.log
