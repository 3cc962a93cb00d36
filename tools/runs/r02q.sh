O=gpurun_out/r02q; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/hit -- python3 tools/pmc_nt.py t256k,q,qp > $O/log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/pmc_nt.py t256k,q,qp >> $O/log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(dict)
for d in ("hit", "fetch"):
    for f in glob.glob("gpurun_out/r02q/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_nt" not in row["Kernel_Name"]:
                continue
            acc[int(row["Dispatch_Id"])]["name"] = row["Kernel_Name"].split("(")[0][-48:]
            acc[int(row["Dispatch_Id"])][row["Counter_Name"] + "@" + d] = float(row["Counter_Value"])
for k in sorted(acc):
    print(k, acc[k])
PY
