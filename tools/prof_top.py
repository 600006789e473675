"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/prof_top.py <output dir> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:rows]:
    print("%9.1f us x%5s %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), r["Name"][:120]))
