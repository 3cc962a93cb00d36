O=gpurun_out/r04u; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "tap" > $O/test_tap.txt 2>&1; tail -2 $O/test_tap.txt
for i in 1 2; do
timeout 600 python -c "import mvp_pytorch_amd.hip as h; h.TAP_MAX=0; import runpy,sys; sys.argv=['bench.py','--steps','20','--warmup','5','--no-extras']; runpy.run_path('bench.py', run_name='__main__')" > $O/bench_old_$i.txt 2>&1; echo "old $(grep -o '"ms_per_step": [0-9.]*' $O/bench_old_$i.txt | head -1)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_new_$i.txt 2>&1; echo "new $(grep -o '"ms_per_step": [0-9.]*' $O/bench_new_$i.txt | head -1)"
done
