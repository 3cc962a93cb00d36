# Round 5: attention backward with the resident workgroups of a CU started out of phase (dense batches: equal lengths run in step)
# (record of a finished experiment: the stagger knob of attn_bwd_fused_kernel was removed after this run — profiles/r05_experiments.txt section 12)
O=gpurun_out/r05ag; mkdir -p $O
cd $GRAFT_REPO_ROOT
export MVPTR_LIB=diag
for st in 0 2 4 6; do
  export MVPTR_NT_EXP=$(( st << 26 ))
  echo "== stagger $st x s_sleep(127) per slot"
  python3 tools/bench_attn.py 2>/dev/null | grep "B=" | cut -c1-200
done
for r in 1 2; do
for st in 0 4; do
  export MVPTR_NT_EXP=$(( st << 26 ))
  python3 bench.py --steps 20 --warmup 6 --no-extras --fixed-length 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('fixed stagger $st', d['ms_per_step'])"
done
done
