# Round 5: tile-height rule of gemm_nt8 in the step (same box, alternating): rule / 256-row tiles only (MVPTR_NT_EXP bits 22-25 = 8) / 224 / round-4 kernels
O=gpurun_out/r05f; mkdir -p $O
cd $GRAFT_REPO_ROOT
export MVPTR_LIB=diag
for i in 1 2; do
for v in rule:0 mt8:33554432 mt7:29360128 old:131072; do
  n=${v%%:*}; e=${v##*:}
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_${n}_$i.log 2>&1; echo -n "$n $i: "; tail -1 $O/bench_${n}_$i.log | cut -c150-200
done
done
for v in rule:0 mt8:33554432; do
  n=${v%%:*}; e=${v##*:}
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length > $O/bench_fixed_${n}.log 2>&1; echo -n "fixed $n: "; tail -1 $O/bench_fixed_${n}.log | cut -c150-200
  MVPTR_NT_EXP=$e python3 bench.py --no-extras --no-cpu-baseline --steps 20 --one-stream > $O/bench_one_${n}.log 2>&1; echo -n "one-stream $n: "; tail -1 $O/bench_one_${n}.log | cut -c150-200
done
unset MVPTR_LIB
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_packed_under_rocprof.log 2>&1
python3 tools/kstats.py $O/packed/packed_kernel_stats.csv 13 2>/dev/null | head -40
find $O -name "*kernel_trace.csv" -size +2M -delete
python3 tools/stash_soak.py --steps 2000 > $O/stash_soak.log 2>&1; tail -5 $O/stash_soak.log
