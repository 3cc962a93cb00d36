O=gpurun_out/r02bo; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -n "passed\|failed\|rc=\|Error\|^FAILED" $O/gputest.log | tail -6
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | sed "s/^/packed new /" | cut -c1-330 | tee -a $O/bench.log
MVPTR_NT_EXP=65536 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | sed "s/^/packed old /" | cut -c1-330 | tee -a $O/bench.log
done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed new /" | cut -c1-330 | tee -a $O/bench.log
MVPTR_NT_EXP=65536 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fixed-length 2>&1 | tail -1 | sed "s/^/fixed old /" | cut -c1-330 | tee -a $O/bench.log
done
