O=gpurun_out/r02j; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 32 "" 32 ""; do
MVPTR_GEMM_TN=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --fixed-length 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TN=$v fixed', d['ms_per_step'], d['value'])"
done
for v in 32 ""; do
MVPTR_GEMM_TN=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TN=$v packed', d['ms_per_step'], d['value'])"
done
