"""Average value of one PMC counter per kernel from a `rocprofv3 --pmc <COUNTER> --output-format csv -d DIR` run.
Usage: python scripts/pmc_write_size.py DIR   (prints the 14 kernels with the largest totals)"""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(list)
name = "counter"
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|zkmi::|^void ", "", r["Kernel_Name"])
        k = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k)
        acc[k[:70]].append(float(r["Counter_Value"]))
        name = r.get("Counter_Name", name)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(k.ljust(72), str(len(v)).rjust(4), "avg %s" % name, round(sum(v) / len(v), 1))
