cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
python3 -m pytest tests/test_model_gpu.py -q -m gpu -s 2>&1 | grep -E "^PARITY|passed|failed" > $O/parity_values.txt; tail -3 $O/parity_values.txt
python3 tools/stash_soak.py --runs 3 --steps 3000 > $O/stash_soak.log 2>&1; grep -v amdgpu $O/stash_soak.log | cut -c1-330 | tail -10
