O=gpurun_out/r02g; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/sweep_tn_group.py 32,q 0,2,3,4,5,6,7,8,10,14 > $O/sweep_group.log 2>&1; cat $O/sweep_group.log
