O=gpurun_out/r03s; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4
python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("one-pass attention backward", d["ms_per_step"])
PY
MVPTR_LIB=diag MVPTR_ATTN_TWO_PASS=1 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench2_$i.json 2> $O/bench2_$i.err; python - <<PY
import json; d=json.loads(open("$O/bench2_$i.json").read().strip().splitlines()[-1]); print("two-pass (diag build)", d["ms_per_step"])
PY
done
