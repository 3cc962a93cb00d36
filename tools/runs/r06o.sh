cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
python3 tools/epi_ablate.py --ms 37748,10917 > $O/epi_ablate.log 2>&1; grep -v amdgpu $O/epi_ablate.log | grep -E "^gemm|ffn1 fwd GELU  |M =" | cut -c1-420
