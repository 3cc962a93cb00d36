cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
for c in model model_arena; do WARM_SIDE=1 timeout 120 python3 tools/debug_capture2.py $c 2>&1 | grep -v amdgpu | grep -E "capturing|capture ended|replayed|Error|error|Segmentation|Warning" | cut -c1-300 | tail -4; echo "   [$c side]"; done
python3 -X faulthandler -m pytest tests -x -q -m gpu -k "graphed_step" > $O/graph.log 2>&1; grep -v amdgpu $O/graph.log | grep -E "passed|failed|^E  |dropout|Error|Fatal|File \"/root" | tail -30 | cut -c1-500
