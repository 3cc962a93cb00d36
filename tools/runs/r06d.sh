# Round 6, call 4: file-level bisect of the unpad-vs-pad gradient difference; first try of the captured step
cd $GRAFT_REPO_ROOT
for v in bisA bisB bisC; do echo "== $v"; MVPTR_LIB=$v python3 tools/debug_ft_grads.py 2>&1 | grep -v amdgpu | grep -A2 "unpad 1 streams 0 defer 0"; done
python3 -m pytest tests -q -m gpu -k "graphed_step" 2>&1 | grep -v amdgpu | grep -E "^FAILED|passed|failed|^E  |dropout|Error|error" | tail -30
