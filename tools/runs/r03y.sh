O=gpurun_out/r03y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/gputest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["config"]["all_slots_valid"]["ms_per_step"], d["config"]["single_stream_model"]["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
