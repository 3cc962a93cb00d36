O=gpurun_out/r02r; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pd_config" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -12 $O/gputest.log | cut -c1-300
timeout 600 python tools/sweep_gemm_cfg.py t256k,pd 64000,19200,37748 > $O/sweep_nt.log 2>&1; cat $O/sweep_nt.log
