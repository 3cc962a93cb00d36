O=gpurun_out/r04r; mkdir -p $O
for i in 1 2; do
MVPTR_LIB=diag MVPTR_NT_EXP=65536 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --one-stream > $O/bench1s_off_$i.txt 2>&1; echo "one-stream off $(grep -o '"ms_per_step": [0-9.]*' $O/bench1s_off_$i.txt)"
MVPTR_LIB=diag timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --one-stream > $O/bench1s_on_$i.txt 2>&1; echo "one-stream on $(grep -o '"ms_per_step": [0-9.]*' $O/bench1s_on_$i.txt)"
done
