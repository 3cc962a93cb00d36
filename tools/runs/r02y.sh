O=gpurun_out/r02y; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== slab"; timeout 900 python tools/sweep_tn_group.py q 0,2,3,4,5,6,7,8,10,12,14 37748,19200,10917 0 2>&1 | grep "^M=" | tee $O/sweep_slab.log
