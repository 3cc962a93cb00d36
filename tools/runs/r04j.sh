# rest of the GPU suite after the last failure + the round-4 profiles
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "device_side_row_count" > $O/pytest_a.txt 2>&1; echo "pytest rc $?" >> $O/pytest_a.txt; tail -4 $O/pytest_a.txt
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $O/pytest_b.txt 2>&1; echo "pytest rc $?" >> $O/pytest_b.txt; tail -4 $O/pytest_b.txt
bash tools/runs/r04_profile.sh > $O/profile.log 2>&1; tail -60 $O/profile.log
