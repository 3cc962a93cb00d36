O=gpurun_out/r02bi; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for T in 0 25000 50000 1000000; do
MVPTR_WGRAD_ASIDE_MAX_ROWS=$T timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | sed "s/^/packed T=$T /" | cut -c1-330 | tee -a $O/bench.log
done
for T in 0 25000 1000000; do
MVPTR_WGRAD_ASIDE_MAX_ROWS=$T timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed T=$T /" | cut -c1-330 | tee -a $O/bench.log
done
done
