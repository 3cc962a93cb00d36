O=gpurun_out/r02i; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 600 python tools/sweep_gemm_cfg.py t256k,q 64000,19200 > $O/sweep_nt.log 2>&1; cat $O/sweep_nt.log
