# Round 5, final code: cold table against hipBLASLt's plain kernels on the step's shapes (the figure VERDICT r04 #1 asks for)
O=gpurun_out/r05ae; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/blas_table.py --ms 37748,10917,64000 2>/dev/null | cut -c1-100 > $O/table.log; cat $O/table.log
