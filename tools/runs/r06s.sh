cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -8 | cut -c1-300
python3 bench.py --model single --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('single', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['hip_graph'])"
python3 tools/graph_soak.py --steps 1500 > $O/graph_soak.log 2>&1; grep -v amdgpu $O/graph_soak.log | cut -c1-400 | tail -12
