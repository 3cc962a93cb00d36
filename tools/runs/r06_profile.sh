# Round-6 profiles (run on the GPU box through gpurun): kernel summaries of the timed bench region, in-step and replay
# roofline of the dominant kernel, PMC traffic, region-feature path, FETCH_SIZE calibration.  Output: gpurun_out/r06prof/
O=gpurun_out/r06prof; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P="--kernel-trace --stats --output-format csv"
# the timed command as the driver runs it (N = 1: the step replayed as a captured HIP graph) ...
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
# ... and with every launch queued from Python (rounds 1-5; the same kernels)
rocprofv3 $P -d $O/packed_eager -o packed_eager -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-graph > $O/bench_packed_eager_under_rocprof.log 2>&1
rocprofv3 $P -d $O/fixed -o fixed -- python3 bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_under_rocprof.log 2>&1
rocprofv3 $P -d $O/single -o single -- python3 bench.py --steps 10 --warmup 3 --no-extras --model single > $O/bench_single_under_rocprof.log 2>&1
# ONE compute stream: no two kernels of the step overlap, so the per-kernel durations are the kernels' own (in-step figures without the overlap)
rocprofv3 $P -d $O/onestream -o onestream -- python3 bench.py --steps 10 --warmup 3 --no-extras --one-stream > $O/bench_onestream_under_rocprof.log 2>&1
Q="--kernel-trace --output-format csv"
rocprofv3 --pmc FETCH_SIZE $Q -d $O/calib_fetch -- python3 tools/calib_fetch.py > $O/calib.log 2>&1
rocprofv3 --pmc FETCH_SIZE $Q -d $O/pmc_packed_fetch -- python3 tools/prof_dominant.py 2 > $O/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/pmc_packed_write -- python3 tools/prof_dominant.py 2 >> $O/pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE $Q -d $O/pmc_full_fetch -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/pmc_full_write -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
rocprofv3 $P -d $O/mix_packed -o mix_packed -- python3 tools/prof_dominant.py 6 >> $O/pmc.log 2>&1
rocprofv3 $P -d $O/mix_full -o mix_full -- python3 tools/prof_dominant.py 6 full >> $O/pmc.log 2>&1
python3 tools/traffic_json.py $O/r06_dominant_traffic.json $O/calib_fetch $O/pmc_packed_fetch $O/pmc_packed_write $O/pmc_full_fetch $O/pmc_full_write > $O/traffic.log 2>&1
python3 tools/roofline_json.py $O/r06_roofline.json $O/mix_packed/mix_packed_kernel_stats.csv $O/mix_full/mix_full_kernel_stats.csv $O/r06_dominant_traffic.json $O/packed/packed_kernel_stats.csv $O/fixed/fixed_kernel_stats.csv > $O/roofline.log 2>&1
# region-feature path (north_star: coalesced HBM loads of the region features evidenced by rocprof HBM GB/s)
rocprofv3 $P -d $O/feat -o feat -- python3 tools/prof_features.py 8 > $O/feat.log 2>&1
rocprofv3 --pmc FETCH_SIZE $Q -d $O/feat_fetch -- python3 tools/prof_features.py 3 >> $O/feat.log 2>&1
rocprofv3 --pmc WRITE_SIZE $Q -d $O/feat_write -- python3 tools/prof_features.py 3 >> $O/feat.log 2>&1
CAL=$(python3 -c "import json;print(json.load(open('$O/r06_dominant_traffic.json'))['fetch_size_calibration']['global_load_dwordx4'])")
python3 tools/features_json.py $O/r06_region_features.json $O/feat/feat_kernel_stats.csv $O/feat_fetch $O/feat_write $CAL > $O/features.log 2>&1
# MFMA / LDS counters of the launch mix (three separate --pmc passes)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE $Q -d $O/pmc_mfma -- python3 tools/prof_dominant.py 2 >> $O/pmc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE $Q -d $O/pmc_lds -- python3 tools/prof_dominant.py 2 >> $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc_mfma $O/pmc_lds > $O/r06_pmc_mfma_lds.csv 2>> $O/pmc.log
# steady-state launch table and GPU idle time of the last five steps (from the traces, before they are deleted)
python3 tools/gpu_idle.py $O/packed/packed_kernel_trace.csv > $O/r06_step_launches_packed.txt 2>&1
python3 tools/gpu_idle.py $O/fixed/fixed_kernel_trace.csv > $O/r06_step_launches_fixed.txt 2>&1
python3 tools/gpu_idle.py $O/packed_eager/packed_eager_kernel_trace.csv > $O/r06_step_launches_packed_eager.txt 2>&1
# keep only the summaries (the traces are tens of MB)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
python3 bench.py > $O/r06_bench_default_line.txt 2> $O/bench_default.err
ls $O; tail -3 $O/bench_packed_under_rocprof.log | cut -c1-300; cat $O/roofline.log | head -70; cat $O/features.log
