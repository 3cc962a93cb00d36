# Round 5: is the re-rank leg's rate a box effect?  (91 k pairs/s in the profile run's default line against 147 k earlier)
O=gpurun_out/r05s; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline > $O/line1.txt 2>/dev/null; tail -1 $O/line1.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['config']['secondary']['configs3_retrieval_rerank'])[:300])"
python3 bench.py --no-cpu-baseline > $O/line2.txt 2>/dev/null; tail -1 $O/line2.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['config']['secondary']['configs3_retrieval_rerank'])[:300])"
nproc; cat /proc/loadavg
