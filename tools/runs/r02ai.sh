O=gpurun_out/r02ai; mkdir -p $O
cd $GRAFT_REPO_ROOT
for M in 37748 64000 19200 10917; do
timeout 600 python tools/sweep_nt_group.py $M "4,0;8,4;8,3;8,2;12,4;16,4;6,4;4,4;4,3;4,2;8,12;16,1;32,1" 2>&1 | grep "^M=" | grep "ffn1\|ffn2 dgrad\|qkv bias" | tee -a $O/sweep.log
done
