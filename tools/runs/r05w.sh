# Round 5: FETCH_SIZE of the stack launch per problem type (which shapes re-fetch their operands?)
O=gpurun_out/r05w; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
Q="--kernel-trace --output-format csv"
for M in 10917; do
rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch_$M -- python3 tools/prof_tn_shapes.py $M > $O/shapes_$M.log 2>&1
grep "operands" $O/shapes_$M.log
python3 tools/pmc_kernel.py $O/fetch_$M FETCH_SIZE gemm_tn_sk_kernel
done
find $O -name "*counter_collection.csv" -size +8M -delete
