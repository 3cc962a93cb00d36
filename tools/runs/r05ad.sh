# Round 5: GPU idle of the UNTRACED step, estimated on one stream: wall time per step minus the sum of the kernels' own times
# (one stream: no two kernels overlap, and a kernel's duration does not depend on the tracer)
O=gpurun_out/r05ad; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for r in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-extras --one-stream 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('untraced one-stream step', d['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/onestream -o onestream -- python3 bench.py --steps 10 --warmup 3 --no-extras --one-stream > $O/traced.log 2>&1
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/r05ad/onestream/onestream_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / 16
print("sum of kernel time per step on one stream (16 traced steps): %.2f ms" % tot)
for l in open("gpurun_out/r05ad/traced.log"):
    if l.startswith("{"):
        print("traced one-stream step", json.loads(l)["ms_per_step"])
PY
python3 tools/gpu_idle.py $O/onestream/onestream_kernel_trace.csv | head -3
find $O -name "*kernel_trace.csv" -size +2M -delete
