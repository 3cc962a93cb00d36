cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02at
timeout 600 python tools/sweep_gemm_cfg.py w4,t256,t256k,pd 37748,11143 2>&1 | grep "^M=" | tee gpurun_out/r02at/sweep.log
