# Round 5: gemm_nt4_kernel (two staggered 4-wave workgroups per CU): correctness under the op tests, then the cold table
# (record of a finished experiment: the u4 / v4 stagger knob and gemm_nt4_kernel were removed after this run — profiles/r05_experiments.txt section 6)
O=gpurun_out/r05p; mkdir -p $O
cd $GRAFT_REPO_ROOT
export MVPTR_LIB=diag
MVPTR_GEMM_CFG=4 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt" > $O/pytest_nt4.log 2>&1; tail -3 $O/pytest_nt4.log
for st in 0 2 3 4 6; do
  export MVPTR_NT_EXP=$(( st << 26 ))
  echo "== cfg 4 stagger $st (x s_sleep(127))"
  python3 tools/blas_table.py --ms 37748 --ab --cfg 4 --loop-only 2>/dev/null | grep -v "^knob\|diagnostic" | cut -c1-150
done > $O/table.log 2>&1
cat $O/table.log
export MVPTR_NT_EXP=$(( 3 << 26 ))
python3 tools/blas_table.py --ms 10917,64000 --ab --cfg 4 2>/dev/null | grep -v "^knob\|diagnostic" | cut -c1-120
