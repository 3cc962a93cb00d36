cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
echo "== global mode"; CAP_MODE=global timeout 120 python3 tools/debug_capture2.py model 2>&1 | grep -v amdgpu | tail -12 | cut -c1-400
echo "== AMD_LOG_LEVEL=1"; AMD_LOG_LEVEL=1 timeout 120 python3 tools/debug_capture2.py model > $O/log1.txt 2>&1; grep -v amdgpu $O/log1.txt | tail -25 | cut -c1-400
which gdb && (gdb -batch -ex run -ex bt --args python3 tools/debug_capture2.py model 2>&1 | tail -40 | cut -c1-300)
