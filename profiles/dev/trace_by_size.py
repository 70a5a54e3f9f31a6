#!/usr/bin/env python3
"""kernel durations of a rocprofv3 kernel trace grouped by (kernel, grid size): python trace_by_size.py <b_kernel_trace.csv> [filter ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:] or ["knn_tree", "kd_", "locate"]
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if any(f in n for f in flt):
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
        acc[(short, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print("%-30s grid %8d wg %4d  n %3d  min %8.1f med %8.1f max %8.1f us" % (k[0], k[1], k[2], len(v), v[0], v[len(v) // 2], v[-1]))
