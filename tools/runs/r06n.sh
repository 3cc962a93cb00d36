cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
python3 -m pytest tests -q -m gpu -k "graphed_step or captured_vqa" 2>&1 | grep -v amdgpu | grep -E "^FAILED|passed|failed|^E  " | tail -8 | cut -c1-300
for g in "" "--no-graph"; do
  python3 bench.py --model single --no-extras --steps 20 --warmup 5 $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('single $g', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['hip_graph'])"
done
python3 - <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
for ng in (False, True):
    out = bench._secondary_legs(torch.device('cuda:0'), 20, ng)
    v = out['configs4_vqa_step']
    print('vqa no_graph=%s' % ng, v.get('ms_per_step'), v.get('step_frac'), v.get('hip_graph'), v.get('error'))
PY
