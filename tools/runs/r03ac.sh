O=gpurun_out/r03ac; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in "" "--one-stream"; do
python bench.py --steps 20 --warmup 5 --no-extras $v > $O/b_$i$v.json 2> $O/b_$i$v.err; python - <<PY
import json; d=json.loads(open("$O/b_$i$v.json").read().strip().splitlines()[-1]); print("'$v'", d["ms_per_step"])
PY
done; done
