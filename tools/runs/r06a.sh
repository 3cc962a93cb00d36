# Round 6, call 1: new GELU pair (polynomial Phi, one exp), device error word, one-wave-per-row feature cast
O=gpurun_out/r06a; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python3 tools/epi_ablate.py --ms 37748,10917 > $O/epi_ablate.log 2>&1; grep -v amdgpu $O/epi_ablate.log | cut -c1-300
python3 tools/blas_table.py --ms 37748,10917,64000 > $O/blas_table.log 2>&1; grep -v amdgpu $O/blas_table.log | tail -30
python3 bench.py > $O/bench_default.log 2>&1; grep -v amdgpu $O/bench_default.log | tail -1 | cut -c1-3000
