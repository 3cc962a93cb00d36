O=gpurun_out/r02ad; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/clock_tn.py h 2>&1 | grep "^M=" | tee $O/clock_h.log
