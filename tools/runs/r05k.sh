# Round 5: FETCH_SIZE of the weight-gradient stack launch, aligned against flat remainder (diagnostic build, knob bit 11)
O=gpurun_out/r05k; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MVPTR_LIB=diag
Q="--kernel-trace --output-format csv"
export MVPTR_NT_EXP=0
rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch_aligned -- python3 tools/prof_dominant.py 2 > $O/pmc.log 2>&1
export MVPTR_NT_EXP=2048
rocprofv3 --pmc FETCH_SIZE $Q -d $O/fetch_flat -- python3 tools/prof_dominant.py 2 >> $O/pmc.log 2>&1
python3 tools/pmc_kernel.py $O/fetch_aligned FETCH_SIZE gemm_tn_sk_kernel
python3 tools/pmc_kernel.py $O/fetch_flat FETCH_SIZE gemm_tn_sk_kernel
find $O -name "*counter_collection.csv" -size +8M -delete
unset MVPTR_LIB MVPTR_NT_EXP
python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"
python3 bench.py --steps 30 --warmup 8 --no-extras --model single 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"
