# Round-2 profiles (run on the GPU box through gpurun): kernel summaries of the timed bench region,
# PMC traffic of the dominant-kernel launch mix, FETCH_SIZE calibration.  Output: gpurun_out/r02p/
O=gpurun_out/r02p; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
rocprofv3 $P -d $O/fixed -o fixed -- python3 bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_under_rocprof.log 2>&1
rocprofv3 $P -d $O/single -o single -- python3 bench.py --steps 10 --warmup 3 --no-extras --model single > $O/bench_single_under_rocprof.log 2>&1
Q="--kernel-trace --output-format csv"
rocprofv3 --pmc FETCH_SIZE $Q -d $O/calib_fetch -- python3 tools/calib_fetch.py > $O/calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/calib_write -- python3 tools/calib_fetch.py >> $O/calib.log 2>&1
rocprofv3 --pmc FETCH_SIZE $Q -d $O/pmc_packed_fetch -- python3 tools/prof_dominant.py 2 > $O/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/pmc_packed_write -- python3 tools/prof_dominant.py 2 >> $O/pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE $Q -d $O/pmc_full_fetch -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/pmc_full_write -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
rocprofv3 $P -d $O/mix_packed -o mix_packed -- python3 tools/prof_dominant.py 6 >> $O/pmc.log 2>&1
rocprofv3 $P -d $O/mix_full -o mix_full -- python3 tools/prof_dominant.py 6 full >> $O/pmc.log 2>&1
python3 tools/traffic_json.py $O/r02_dominant_traffic.json $O/calib_fetch $O/pmc_packed_fetch $O/pmc_packed_write $O/pmc_full_fetch $O/pmc_full_write > $O/traffic.log 2>&1
python3 tools/roofline_json.py $O/r02_roofline.json $O/mix_packed/mix_packed_kernel_stats.csv $O/mix_full/mix_full_kernel_stats.csv $O/r02_dominant_traffic.json > $O/roofline.log 2>&1
# keep only the summaries (the traces are tens of MB)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls -la $O $O/*; tail -3 $O/bench_packed_under_rocprof.log | cut -c1-300; cat $O/traffic.log | head -60
