# Round 5: trial — LayerNorm fold in EVERY no-grad eval forward: which tests notice?
O=gpurun_out/r05v; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|Error" $O/pytest_gpu.log | tail -30
