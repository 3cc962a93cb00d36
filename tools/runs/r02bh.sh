O=gpurun_out/r02bh; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed new /" | cut -c1-330 | tee -a $O/bench.log
MVPTR_WGRAD_ASIDE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed old /" | cut -c1-330 | tee -a $O/bench.log
done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --model single 2>&1 | tail -1 | sed "s/^/single new /" | cut -c1-330 | tee -a $O/bench.log
MVPTR_WGRAD_ASIDE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --model single 2>&1 | tail -1 | sed "s/^/single old /" | cut -c1-330 | tee -a $O/bench.log
done
