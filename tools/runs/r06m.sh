cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -12 | cut -c1-400
python3 bench.py > $O/bench_default.log 2>&1; grep -v amdgpu $O/bench_default.log | tail -1 | cut -c1-3500
for i in 1 2; do
  python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('graph   ', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['hip_graph'])"
  python3 bench.py --no-extras --steps 20 --warmup 5 --no-graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no graph', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"
done
