cd $GRAFT_REPO_ROOT
for args in "salt" "fwd" "fwd --one-stream" "fwd --no-wra" "fwd --one-stream --no-wra" "fwdbwd --one-stream --no-wra" "fwdbwd --one-stream" "fwdbwd" "clip" "full"; do
  timeout 120 python3 tools/debug_capture.py $args 2>&1 | grep -v amdgpu | grep -E "capturing|capture ended|replayed|Error|error|Segmentation" | cut -c1-300 | tail -4; echo "   rc=$? [$args]"
done
