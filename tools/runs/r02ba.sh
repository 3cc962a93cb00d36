cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ba
for M in 37748 64000 10917; do
timeout 600 python tools/sweep_nt_group.py $M "6,4;6,4x;4,4x;4,0x;8,0x;16,0x;1,0x;2,12x;4,3;4,3x;8,3x;16,12x" 2>&1 | grep "^M=" | grep "ffn1\|ffn2 dgrad\|qkv bias" | tee -a gpurun_out/r02ba/sweep.log
done
