O=gpurun_out/r03m; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wra" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "packed_pipeline or stream_placement or cfg1" 2>&1 | tail -15
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("packed+wra kernel", d["ms_per_step"])
PY
done
