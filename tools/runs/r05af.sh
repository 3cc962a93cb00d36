# Round 5: cheaper dithered stash code (zero point 27, byte-1 extraction): full suite, cold FFN1, step
O=gpurun_out/r05af; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/pytest_gpu.log | tail -5
python3 tools/blas_table.py --ms 37748,64000 2>/dev/null | grep "ffn1 fwd\|sum" | cut -c1-100
for r in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'], d['value'])"; done
python3 tools/stash_soak.py --runs 4 --steps 3000 > $O/soak.log 2>&1; grep -v amdgpu $O/soak.log | cut -c1-300 | tail -3
