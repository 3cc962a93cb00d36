O=gpurun_out/r03u; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
python3 tools/kstats.py $O/packed/packed_kernel_stats.csv 13 200 > $O/kstats.txt; head -3 $O/kstats.txt | cut -c1-200; sed -n "3,30p" $O/kstats.txt | cut -c1-150
python3 tools/gpu_idle.py $O/packed/packed_kernel_trace.csv > $O/gpu_idle.txt 2>&1; sed -n "1,2p;40,90p" $O/gpu_idle.txt | cut -c1-150
find $O -name "*kernel_trace.csv" -size +2M -delete
