# Round 5, final code: full GPU suite with the parity prints, smoke, then the round's profiles and the default bench line
O=gpurun_out/r05final; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -3
grep -h "^PARITY\|dropout replay\|configs\[4\] gradient\|stash u8\|LN fold\|LayerNorm-folded" $O/pytest_gpu.log > $O/parity_values.txt; wc -l $O/parity_values.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/runs/r05_profile.sh > $O/profile_script.log 2>&1
cp gpurun_out/r05prof/r05_bench_default_line.txt $O/
tail -1 $O/r05_bench_default_line.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['second_kernel']['frac'], d['config']['all_slots_valid']['ms_per_step'])"
