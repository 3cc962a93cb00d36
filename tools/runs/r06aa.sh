cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
python3 -m pytest tests -q -m gpu -k "probs_tensor or hidden_states_attentions or fall_back_when_memory" > $O/new.log 2>&1; grep -v amdgpu $O/new.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -12 | cut -c1-400
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -6 | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300
