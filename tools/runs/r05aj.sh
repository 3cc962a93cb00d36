# (round-5 diagnostic run; see profiles/r05_experiments.txt section 13)
cd $GRAFT_REPO_ROOT
python3 tools/tap_debug.py 2>/dev/null | grep tap_rows
