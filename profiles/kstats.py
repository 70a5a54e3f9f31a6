"""Print a rocprofv3 b_kernel_stats.csv compactly: python profiles/kstats.py gpurun_out/t1/b_kernel_stats.csv"""
import csv
import re
import sys

tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:64]
    print(f"{n:64s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.2f} us {r['Percentage']:>6s}%")
