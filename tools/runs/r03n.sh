O=gpurun_out/r03n; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "inputs_ready or packed_pipeline or stream_placement or stager or pipeline" 2>&1 | tail -15
for i in 1 2 3; do
for v in "" "--no-ready-event"; do
python bench.py --steps 20 --warmup 5 --no-extras $v > $O/bench_$i$v.json 2> $O/bench_$i$v.err; python - <<PY
import json; d=json.loads(open("$O/bench_$i$v.json").read().strip().splitlines()[-1]); print("packed '$v'", d["ms_per_step"])
PY
done; done
python3 tools/cpu_enqueue.py > $O/cpu_enqueue.txt 2>&1; sed -n "3,6p" $O/cpu_enqueue.txt | cut -c1-180
