O=gpurun_out/r02bj; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|^FAILED" $O/gputest.log | tail -8
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | sed "s/^/packed /" | cut -c1-330 | tee -a $O/bench.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed /" | cut -c1-330 | tee -a $O/bench.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --model single 2>&1 | tail -1 | sed "s/^/single /" | cut -c1-330 | tee -a $O/bench.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
