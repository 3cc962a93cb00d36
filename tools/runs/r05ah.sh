# (round-5 diagnostic run; see profiles/r05_experiments.txt section 13)
# Round 5: launch sequence of one step on one stream (which small kernels sit between the encoder stacks?)
O=gpurun_out/r05ah; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 3 --warmup 3 --no-extras --one-stream > $O/run.log 2>&1
python3 tools/step_sequence.py $O/t/t_kernel_trace.csv > $O/sequence.txt 2>&1; tail -3 $O/sequence.txt; wc -l $O/sequence.txt

cp $O/t/t_kernel_trace.csv $O/trace_small.csv 2>/dev/null; ls -la $O/t/ | head -5
