cd $GRAFT_REPO_ROOT
for c in model_loss1 model_loss2 model_loss3 model_loss4 model_loss5; do
  timeout 120 python3 tools/debug_capture2.py $c 2>&1 | grep -v amdgpu | grep -E "capturing|capture ended|replayed|Error|error|Segmentation" | cut -c1-300 | tail -3; echo "   [$c]"
done
