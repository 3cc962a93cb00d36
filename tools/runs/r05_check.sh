# Round 5: sanity of the committed tree — GPU suite, smoke, one short bench line
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed" | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c1-160
