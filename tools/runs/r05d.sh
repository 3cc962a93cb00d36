# Round 5: ping-pong loop experiment (gemm_nt8_kernel, diagnostic build, MVPTR_GEMM_CFG=8): parity of every epilogue, then the cold table with loop-only columns
O=gpurun_out/r05d; mkdir -p $O
cd $GRAFT_REPO_ROOT
MVPTR_LIB=diag MVPTR_GEMM_CFG=8 python3 -m pytest tests/test_ops_gpu.py -x -q -k "gemm_nt" > $O/pytest_nt8.log 2>&1; tail -3 $O/pytest_nt8.log
MVPTR_LIB=diag python3 tools/blas_table.py --ab --cfg 8 --loop-only --ms 37748,10917,64000 --reps 4 > $O/table.log 2>&1; cat $O/table.log
