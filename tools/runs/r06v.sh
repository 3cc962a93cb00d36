cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06v; mkdir -p $O
ORD="6,4;4,4;4,0;2,0;8,4;8,2;16,2;32,1;3,12;4,6;12,3"
python3 tools/sweep_nt_group.py 37748 "$ORD" 2>&1 | grep -v amdgpu | cut -c1-400
Q="--kernel-trace --output-format csv"
PMC_ONCE=1 rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch -- python3 tools/sweep_nt_group.py 37748 "$ORD" > $O/fetch.log 2>&1
for k in "gemm_nt8_kernel<1, 8>" "gemm_nt8_kernel<0, 8>" "gemm_nt8_kernel<3, 8>" "gemm_nt8_kernel<2, 8>"; do python3 tools/pmc_kernel.py $O/fetch FETCH_SIZE "$k" | cut -c1-400; done
# LayerNorm kernels: bytes moved per launch against the algorithmic 2 + 2 B per element (NS-1: would statistics from the GEMM save a byte?)
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE $Q -d $O/ln -- python3 tools/bench_ln.py > $O/ln.log 2>&1
python3 tools/pmc_kernel.py $O/ln FETCH_SIZE ln_fwd_j | cut -c1-300; python3 tools/pmc_kernel.py $O/ln WRITE_SIZE ln_fwd_j | cut -c1-300
python3 tools/pmc_kernel.py $O/ln FETCH_SIZE ln_bwd_j | cut -c1-300; python3 tools/pmc_kernel.py $O/ln WRITE_SIZE ln_bwd_j | cut -c1-300
grep -v amdgpu $O/ln.log | grep "M=37748" | head -3 | cut -c1-300
find $O -name "*counter_collection.csv" -size +4M -delete
