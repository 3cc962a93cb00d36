O=gpurun_out/r03w; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/gputest.log | tail -3
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("one-allocation arena", d["ms_per_step"])
PY
done
