# (round-5 diagnostic run; see profiles/r05_experiments.txt section 13)
O=gpurun_out/r05ai; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/bench_tap.py 2>/dev/null | grep bound
