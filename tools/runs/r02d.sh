O=gpurun_out/r02d; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "tiny_bi_hn or b64_vs_oracle or b64_decode_flags" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
timeout 1500 python tests/diag_cfg1_grads.py 64 > $O/diag64.log 2>&1
timeout 1500 python tests/diag_cfg1_grads.py 256 > $O/diag256.log 2>&1
grep -E "FAILED|passed|failed" $O/gputest.log | tail; tail -32 $O/diag64.log; tail -32 $O/diag256.log
