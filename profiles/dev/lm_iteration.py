#!/usr/bin/env python3
"""composition of ONE LM iteration from a rocprofv3 kernel trace (between two k_lin_fused of a solve): python lm_iteration.py <trace.csv> [which]
which: index of the iteration from the end (default 3)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:26]
idx = [i for i, r in enumerate(rows) if "k_lin_fused" in r["Kernel_Name"] or "k_lin_surfel" in r["Kernel_Name"]]
pairs = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any("k_chol_step" in rows[j]["Kernel_Name"] or "k_schur_form" in rows[j]["Kernel_Name"] for j in range(a, b))]
a, b = pairs[-back]
t0, prev = int(rows[a]["Start_Timestamp"]), int(rows[a]["Start_Timestamp"])
tot = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    t = tot.setdefault(n, [0, 0.0, 0.0])
    t[0] += 1; t[1] += (e - s) / 1e3; t[2] += max(0, s - prev) / 1e3
    prev = e
print("iteration span %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
if len(sys.argv) > 3:  # every launch of the iteration: start offset, duration, workgroups
    for r in rows[a:b]:
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
        print("  %8.1f  %6.1f us  %5d wg  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, gx, short(r["Kernel_Name"])))
for n, (c, d, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%3d  busy %7.1f us  idle before %6.1f us" % (n, c, d, g))
