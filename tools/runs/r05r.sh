# Round 5: tile order of gemm_nt8 (MVPTR_NT_GROUP sweep; the defaults were tuned for the round-2 kernel)
O=gpurun_out/r05r; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/sweep_nt_group.py 37748 "0,0;4,4;6,4;8,4;2,4;4,3;8,3;4,6;8,6;4,12;8,12;16,12;16,4" 2>/dev/null | grep "^M=" > $O/sweep.log
python3 tools/sweep_nt_group.py 11143 "0,0;4,4;8,4;2,4;4,3;4,6;4,12;8,12;16,4" 2>/dev/null | grep "^M=" >> $O/sweep.log
cat $O/sweep.log
