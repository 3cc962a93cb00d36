O=gpurun_out/r02k; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "qp_config" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -25 $O/gputest.log | cut -c1-300
timeout 600 python tools/ablate_ntq.py > $O/ablate.log 2>&1; cat $O/ablate.log
