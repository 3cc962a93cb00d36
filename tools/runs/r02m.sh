O=gpurun_out/r02m; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_tn" 2>&1 | tail -2
timeout 600 python tools/sweep_tn_group.py q 0 64000,37748,19200,10917 2>&1 | tee $O/sweep_group.log
