cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02az
timeout 900 python tools/soak.py 300 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02az/soak.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
