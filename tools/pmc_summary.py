"""Average rocprofv3 --pmc counter values per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if sub in k:
        print(k[:100])
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
