O=gpurun_out/r04d; mkdir -p $O
timeout 300 python3 tools/debug_ntp.py > $O/debug.txt 2>&1; grep "differ" $O/debug.txt
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "gemm_nt" > $O/pytest_ops.txt 2>&1; echo "pytest rc $?" >> $O/pytest_ops.txt
tail -5 $O/pytest_ops.txt
MVPTR_LIB=diag timeout 600 python3 tools/blas_table.py --ab --ms 10917,37748,64000 > $O/blas_table_ab.txt 2>&1
cat $O/blas_table_ab.txt
