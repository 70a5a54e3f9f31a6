#!/usr/bin/env python3
"""kernels of ONE facade sweep in front of its solve (undistortion, extraction, pose updates, search, window build) from a rocprofv3 kernel
trace of profiles/time_facade.py: python facade_timeline.py <b_kernel_trace.csv> [sweeps back from the end, default 1]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
feat = [i for i, r in enumerate(rows) if "k_features" in r["Kernel_Name"]]
i_feat = feat[-back]
# back to the previous sweep's last LM kernel
i0 = i_feat
while i0 > 0 and not any(s in rows[i0 - 1]["Kernel_Name"] for s in ("k_lm_step", "k_chol_back", "k_gather", "k_lin_fused")):
    i0 -= 1
i1 = i_feat
while i1 < len(rows) and "k_pcr_init" not in rows[i1]["Kernel_Name"]:
    i1 += 1
t0, prev = int(rows[i0]["Start_Timestamp"]), int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "rocprim" in n:
        n = "rocprim::" + ("onesweep" if "onesweep" in n else "histogram" if "histogram" in n else "scan" if "scan" in n else "other")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-32s %8.1f -> %8.1f  (%6.1f us, gap %6.1f) grid %s" % (n[:32], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Grid_Size_X"]))
    prev = e
