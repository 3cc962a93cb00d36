O=gpurun_out/r03l; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
python3 tools/gpu_idle.py $O/packed/packed_kernel_trace.csv > $O/gpu_idle.txt 2>&1; sed -n "1,3p;40,90p" $O/gpu_idle.txt | cut -c1-220
python3 tools/cpu_enqueue.py > $O/cpu_enqueue.txt 2>&1; sed -n "1,80p" $O/cpu_enqueue.txt | cut -c1-180
find $O -name "*kernel_trace.csv" -size +2M -delete
