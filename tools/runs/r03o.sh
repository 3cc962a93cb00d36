O=gpurun_out/r03o; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/cpu_enqueue.py > $O/cpu_enqueue.txt 2>&1; grep -n "python-level tensor" -A72 $O/cpu_enqueue.txt | cut -c1-170
