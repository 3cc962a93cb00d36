O=gpurun_out/r02t; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --steps 8 --warmup 3 --no-extras > $O/log 2>&1
python3 tools/gpu_idle.py $O/tr/tr_kernel_trace.csv
rocprofv3 --kernel-trace --output-format csv -d $O/trf -o trf -- python3 bench.py --steps 8 --warmup 3 --no-extras --fixed-length > $O/logf 2>&1
python3 tools/gpu_idle.py $O/trf/trf_kernel_trace.csv
rm -f $O/tr/*trace.csv $O/trf/*trace.csv
