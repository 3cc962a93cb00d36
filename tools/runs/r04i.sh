# sync-free step A/B on one box + targeted tests + the 2-workgroups-per-CU tile (v4) against the default
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "device_side_row_count or sync_free or packed_training_path or packed_pipeline" > $O/pytest_new.txt 2>&1; echo "pytest rc $?" >> $O/pytest_new.txt
tail -8 $O/pytest_new.txt; grep PARITY $O/pytest_new.txt
for i in 1 2; do
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras > $O/free_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/free_$i.json'));print('sync-free', d['ms_per_step'])"
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --count-readbacks > $O/rb_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/rb_$i.json'));print('read-backs', d['ms_per_step'])"
done
MVPTR_LIB=diag timeout 600 python3 tools/blas_table.py --ab --cfg v4 --ms 10917,37748 > $O/blas_v4.txt 2>&1; cat $O/blas_v4.txt
