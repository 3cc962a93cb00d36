# Round 5: GPU tests touched by the DP / parity / head changes, then the default bench line (all legs) on this box
O=gpurun_out/r05h; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -x -q -s -k "dp or two_rank or rccl or launcher or configs4 or stash or finetune or half or branches" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
grep -h "PARITY\|configs\[4\] gradient\|stash u8" $O/pytest.log | head -20
python3 bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default_line.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05h/bench_default_line.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('step_frac'), d['roofline']['second_kernel']['frac'])
c=d['config']
for k in ('all_slots_valid','single_stream','with_input_pipeline','secondary'):
    print(k, json.dumps(c.get(k))[:700])
print(json.dumps(d.get('cpu_baseline'))[:300])
PY
