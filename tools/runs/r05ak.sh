# (round-5 diagnostic run; see profiles/r05_experiments.txt section 13)
# Round 5: tap_sum with the rows of four entries in flight together (a row gathered 87 times was one wave's serial 200 us)
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu -k "tap or packed_pipeline or packed_training or pretrain_step" 2>&1 | grep "passed\|failed" | tail -2
python3 tools/bench_tap.py 2>/dev/null | grep bound
python3 tools/tap_debug.py 2>/dev/null | grep tap_rows | cut -c1-330
for r in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'], d['value'])"; done
