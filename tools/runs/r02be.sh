O=gpurun_out/r02be; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for L in 2 1 0; do
MVPTR_HEADS_BESIDE=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | sed "s/^/L$L /" | cut -c1-330 | tee -a $O/bench.log
done
done
for L in 2 1 0; do
MVPTR_HEADS_BESIDE=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed L$L /" | cut -c1-330 | tee -a $O/bench.log
done
