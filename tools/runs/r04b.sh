# round 4: persistent gemm_nt kernel + 8-bit gelu' stash: op tests, A/B table against the round-3 kernel and hipBLASLt
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "gemm_nt" > $O/pytest_ops.txt 2>&1; echo "pytest rc $?" >> $O/pytest_ops.txt
tail -15 $O/pytest_ops.txt
MVPTR_LIB=diag timeout 600 python3 tools/blas_table.py --ab --ms 10917,37748,64000 > $O/blas_table_ab.txt 2>&1
cat $O/blas_table_ab.txt
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-400 $O/bench.json
