O=gpurun_out/r03v; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python bench.py --with-input-pipeline --no-extras --steps 20 --warmup 5 > $O/bench_pipeline.json 2> $O/bench_pipeline.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value","ms_per_step","roofline","cpu_baseline")}, indent=None)[:1500])
print(json.dumps(d["config"])[:1200])
p=json.loads(open("$O/bench_pipeline.json").read().strip().splitlines()[-1])
print("pipeline leg", json.dumps(p.get("with_input_pipeline") or p["config"].get("with_input_pipeline"))[:600], p["ms_per_step"])
PY
