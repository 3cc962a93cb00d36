O=gpurun_out/r03z; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_tn" > $O/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error" $O/gputest.log | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu > $O/gputest2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|Error" $O/gputest2.log | tail -5
for i in 1 2 3; do
MVPTR_LIB=diag python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("slabs for few-row weight gradients", d["ms_per_step"])
PY
MVPTR_LIB=diag MVPTR_NT_EXP=4096 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench2_$i.json 2> $O/bench2_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench2_$i.json").read().strip().splitlines()[-1]); print("atomics everywhere", d["ms_per_step"])
PY
done
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("avg_launch_us_back_to_back"))
PY
