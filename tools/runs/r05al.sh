# (round-5 diagnostic run; see profiles/r05_experiments.txt section 13)
# Round 5: tap_rows_bwd by counting sort against the linked lists of round 4 (two builds, same box, alternating)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for tag in list sort; do
  MVPTR_LIB=$tag python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', d['ms_per_step'])"
done
done
