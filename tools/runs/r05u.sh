# Round 5: LayerNorm folded into the neighbouring GEMMs (inference path): tests, re-rank leg with / without
O=gpurun_out/r05u; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -s -k "gemm_nt_ln or folded or retriev or rerank or configs3" > $O/pytest_fold.log 2>&1; grep -n "passed\|failed\|LN fold\|LayerNorm-folded\|Error\|error" $O/pytest_fold.log | tail -20
