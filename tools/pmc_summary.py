#!/usr/bin/env python
"""Mean counter value per kernel from rocprofv3 --pmc CSV output directories."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%s,%s,%s,%d,%.6g" % (d.rstrip("/").split("/")[-1], k, c, len(v), sum(v) / len(v)))
