O=gpurun_out/r02ae; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_tn" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -12 $O/gputest.log | cut -c1-300
timeout 600 python tools/sweep_tn_group.py q,h,o 0 64000,37748,19200,10917 0 2>&1 | grep "^M=" | tee $O/sweep.log
timeout 300 python tools/clock_tn.py o 2>&1 | grep "^M=" | tee $O/clock_o.log
