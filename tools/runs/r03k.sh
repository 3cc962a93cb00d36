O=gpurun_out/r03k; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err; cat $O/bench_default.json | cut -c1-3000
