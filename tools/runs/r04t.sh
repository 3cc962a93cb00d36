O=gpurun_out/r04t; mkdir -p $O
for i in 1 2; do
for v in 65536 0 131072; do
MVPTR_LIB=diag MVPTR_NT_EXP=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --one-stream > $O/os_${v}_$i.txt 2>&1; echo "one-stream exp=$v $(grep -o '"ms_per_step": [0-9.]*' $O/os_${v}_$i.txt | head -1)"
done; done
for v in 65536 0 131072; do
MVPTR_LIB=diag MVPTR_NT_EXP=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/ts_${v}.txt 2>&1; echo "two-stream exp=$v $(grep -o '"ms_per_step": [0-9.]*' $O/ts_${v}.txt | head -1)"
done
