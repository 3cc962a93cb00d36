O=gpurun_out/r05l; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/cpu_enqueue.py > $O/cpu_enqueue.log 2>&1; head -60 $O/cpu_enqueue.log | cut -c1-180
python3 tools/bench_ln.py > $O/ln.log 2>&1; cat $O/ln.log | grep "M="
