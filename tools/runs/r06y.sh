cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06y; mkdir -p $O
python3 tools/kflip_probe.py 37748 2>&1 | grep -v amdgpu | cut -c1-300
Q="--kernel-trace --output-format csv"
PMC_ONCE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch -- python3 tools/kflip_probe.py 37748 > $O/fetch.log 2>&1
for k in "gemm_nt8_kernel<1, 8>" "gemm_nt8_kernel<0, 8>" "gemm_nt8_kernel<3, 8>" "gemm_nt8_kernel<2, 8>" "gemm_nt8_kernel<4, 8>"; do python3 tools/pmc_kernel.py $O/fetch FETCH_SIZE "$k" | cut -c1-300; done
for i in 1 2 3; do
  MVPTR_LIB=diag python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K forward ', d['ms_per_step'])"
  MVPTR_LIB=diag MVPTR_NT_EXP=64 python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K flipping', d['ms_per_step'])"
done
find $O -name "*counter_collection.csv" -size +4M -delete
