cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bn
for M in 37748 64000 10917; do
timeout 600 python tools/pad_ld.py $M 2>&1 | grep "^M=" | tee -a gpurun_out/r02bn/pad.log
done
