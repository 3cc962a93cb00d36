O=gpurun_out/r02o; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -8 $O/gputest.log | cut -c1-400
for m in bi single; do timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --model $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])"; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --fixed-length 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed', d['ms_per_step'], d['value'])"
