# Round 5: aux rows of the whole tile requested up front in gemm_nt8's epilogue (nt_epilogue AUXD): cold table, op tests, step
O=gpurun_out/r05q; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt" > $O/pytest_nt.log 2>&1; tail -2 $O/pytest_nt.log
python3 tools/blas_table.py --ms 37748,10917 2>/dev/null | cut -c1-110 > $O/table.log; cat $O/table.log
for r in 1 2; do
python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'], d['value'])"
done
python3 bench.py --steps 30 --warmup 8 --no-extras --fixed-length 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('fixed', d['ms_per_step'], d['value'])"
