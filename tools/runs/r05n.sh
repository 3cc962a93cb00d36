# Round 5: start skew between the workgroups of an XCD that share an operand panel (weight-gradient stack launch)
O=gpurun_out/r05n; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MVPTR_LIB=diag
for sk in 0 1 2 4 8; do
  export MVPTR_NT_EXP=$(( (sk + 1) << 26 ))
  echo "== skew $sk (MVPTR_NT_EXP=$MVPTR_NT_EXP)"
  python3 tools/bench_tn_stack.py --reps 8 2>/dev/null | grep "M=" | cut -c1-120
done
Q="--kernel-trace --output-format csv"
for sk in 0 2 4; do
  export MVPTR_NT_EXP=$(( (sk + 1) << 26 ))
  rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch_sk$sk -- python3 tools/prof_dominant.py 2 > $O/pmc.log 2>&1
  python3 tools/pmc_kernel.py $O/fetch_sk$sk FETCH_SIZE gemm_tn_sk_kernel
done
find $O -name "*counter_collection.csv" -size +8M -delete
