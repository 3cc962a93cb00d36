# Round 5: what do the bias-gradient column sums cost inside the weight-gradient stack launch?
O=gpurun_out/r05x; mkdir -p $O
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python3 tools/bench_tn_stack.py --reps 8 2>/dev/null | grep "M=" | cut -c1-110
python3 tools/bench_tn_stack.py --reps 8 --no-colsum 2>/dev/null | grep "M=" | cut -c1-110
done
