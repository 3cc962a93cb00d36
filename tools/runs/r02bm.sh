O=gpurun_out/r02bm; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest_$i.log 2>&1; echo "run $i rc=$?"; grep "passed\|failed" $O/gputest_$i.log | tail -1
done
MVPTR_WGRAD_ASIDE=1 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest_aside.log 2>&1; echo "aside rc=$?"; grep "passed\|failed" $O/gputest_aside.log | tail -1
MVPTR_TN_SLAB=1 MVPTR_GEMM_TN=o timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest_slab_o.log 2>&1; echo "slab+o rc=$?"; grep "passed\|failed" $O/gputest_slab_o.log | tail -1
