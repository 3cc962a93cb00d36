# Round 5: full GPU suite (incl. the new tests), then the round's profiles
O=gpurun_out/r05i; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -3
grep -h "^PARITY\|dropout replay\|configs\[4\] gradient\|stash u8" $O/pytest_gpu.log > $O/parity_values.txt; tail -8 $O/parity_values.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
