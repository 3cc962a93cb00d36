O=gpurun_out/r02w; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
python3 tools/gpu_idle.py $(find $O/tr -name "*kernel_trace.csv" | head -1) | cut -c1-220 | tee $O/idle.log
find $O -name "*kernel_trace.csv" -size +2M -delete
