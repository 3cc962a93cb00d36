# sync-free step + new kernels: targeted tests first, then the whole GPU suite, then the bench
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "device_side_row_count or sync_free or hard_negative_mine or bce_logits or packed_training_path or packed_pipeline or b64" > $O/pytest_new.txt 2>&1; echo "pytest rc $?" >> $O/pytest_new.txt
tail -30 $O/pytest_new.txt
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-700 $O/bench.json
