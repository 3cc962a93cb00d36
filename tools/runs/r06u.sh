cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06u; mkdir -p $O
Q="--kernel-trace --output-format csv"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY $Q -d $O/p1 -- python3 tools/pmc_epi.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD $Q -d $O/p2 -- python3 tools/pmc_epi.py > $O/p2.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM $Q -d $O/p3 -- python3 tools/pmc_epi.py > $O/p3.log 2>&1
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY; do for k in "ILi0ELi8" "ILi1ELi8" "ILi7ELi8" "ILi3ELi8"; do python3 tools/pmc_kernel.py $O/p1 $c "gemm_nt8_kernel<${k:3:1}, 8>" 2>/dev/null | cut -c1-150; done; done
for c in SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD; do for k in 0 1 7 3; do python3 tools/pmc_kernel.py $O/p2 $c "gemm_nt8_kernel<$k, 8>" 2>/dev/null | cut -c1-150; done; done
for c in SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM; do for k in 0 1 7 3; do python3 tools/pmc_kernel.py $O/p3 $c "gemm_nt8_kernel<$k, 8>" 2>/dev/null | cut -c1-150; done; done
tail -3 $O/p3.log | cut -c1-200
find $O -name "*counter_collection.csv" -size +4M -delete
