O=gpurun_out/r04ab; mkdir -p $O
MVPTR_BENCH_DEVICE=0 MVPTR_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 40 --warmup 5 --batch 64 --no-extras > $O/two_ranks.txt 2>&1; tail -1 $O/two_ranks.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['value'], json.dumps(d['config']['data_parallel']))"
