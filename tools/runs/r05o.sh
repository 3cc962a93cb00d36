# Round 5: two 4-wave workgroups per CU (128 x 256 and 256 x 128 tiles, BK 32 ring) with the second workgroup of every CU started late
# (record of a finished experiment: the u4 / v4 stagger knob and gemm_nt4_kernel were removed after this run — profiles/r05_experiments.txt section 6)
O=gpurun_out/r05o; mkdir -p $O
cd $GRAFT_REPO_ROOT
export MVPTR_LIB=diag
for cfg in u4 v4; do
for st in 0 3 6 9 22 25; do
  export MVPTR_NT_EXP=$(( st << 26 ))
  echo "== cfg $cfg stagger $st (x s_sleep(127); +16 = odd local index instead of blockIdx >= 256)"
  python3 tools/blas_table.py --ms 37748 --ab --cfg $cfg 2>/dev/null | grep -v "^knob\|diagnostic" | cut -c1-110
done
done > $O/table.log 2>&1
cat $O/table.log
