cd $GRAFT_REPO_ROOT
for c in torch enc_fwdonly enc enc_nodefer enc_arena tap model model_nodefer model_arena; do
  timeout 120 python3 tools/debug_capture2.py $c 2>&1 | grep -v amdgpu | grep -E "capturing|capture ended|replayed|Error|error|Segmentation" | cut -c1-300 | tail -3; echo "   [$c]"
done
