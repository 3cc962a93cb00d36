# Round 5: full GPU suite, smoke, default bench line (after the re-rank read-back change)
O=gpurun_out/r05t; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py > $O/r05_bench_default_line.txt 2> $O/bench.err; tail -1 $O/r05_bench_default_line.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['second_kernel']['frac'], d['config']['all_slots_valid']['ms_per_step'], json.dumps(d['config']['secondary']['configs3_retrieval_rerank'])[:200])"
cat /proc/loadavg
