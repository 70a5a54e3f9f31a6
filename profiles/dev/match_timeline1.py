#!/usr/bin/env python3
"""timeline of the LAST wc_match of a rocprofv3 kernel trace (one search: from its k_features to its k_emit_pairs): python match_timeline1.py <b_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if "k_features" in r["Kernel_Name"]][-1]
t0, prev = int(rows[i0]["Start_Timestamp"]), int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "rocprim" in n:
        n = "rocprim::" + ("onesweep" if "onesweep" in n else "histogram" if "histogram" in n else "scan" if "scan" in n else "other")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s %8.1f -> %8.1f  (%6.1f us, gap %5.1f) grid %s" % (n[:30], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Grid_Size_X"]))
    prev = e
