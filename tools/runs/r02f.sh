O=gpurun_out/r02f; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm_tn" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 600 python tools/sweep_tn.py auto,32,K,q > $O/sweep_tn.log 2>&1; cat $O/sweep_tn.log
