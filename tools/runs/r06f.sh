cd $GRAFT_REPO_ROOT
python3 tools/debug_ft_grads.py 2>&1 | grep -v amdgpu | grep -A3 "unpad 1 streams 1 defer 1\|unpad 1 streams 0 defer 0"
python3 -m pytest tests -q -m gpu -k "graphed_step" 2>&1 | grep -v amdgpu | grep -E "^FAILED|passed|failed|^E  |dropout|Error|error" | cut -c1-900 | tail -20
