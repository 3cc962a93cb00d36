cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
python3 tools/epi_ablate.py --ms 37748,10917 > $O/epi_ablate.log 2>&1; grep -v amdgpu $O/epi_ablate.log | grep -E "^gemm|ffn1 fwd GELU|M =" | cut -c1-250
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -8 | cut -c1-300
for i in 1 2 3; do python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'])"; done
python3 bench.py --no-extras --steps 20 --warmup 5 --fixed-length 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fixed', d['ms_per_step'], d['roofline']['step_frac'])"
