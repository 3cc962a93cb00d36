cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -6 | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300
python3 bench.py > $O/bench.log 2>&1; grep -v amdgpu $O/bench.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'all_slots', d['config']['all_slots_valid']['ms_per_step'], d['config']['all_slots_valid']['step_frac'], 'host', d['config']['host_enqueue_ms_per_step'], d['config']['hip_graph'])
print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'achieved', 'peak', 'frac', 'traffic', 'step_frac')})
print('single', d['config']['single_stream_model'].get('ms_per_step'), 'vqa', d['config']['secondary']['configs4_vqa_step'].get('ms_per_step'))
print('cpu', d.get('cpu_baseline'))
"
