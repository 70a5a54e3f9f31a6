#!/usr/bin/env python3
"""timeline of ONE wc_match_pair of the step-like window from a rocprofv3 kernel trace: python match_timeline.py <b_kernel_trace.csv>
(the trace of `python profiles/dev/time_match.py pair`): per stream, kernel start / end relative to the call's first kernel"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the last k_emit_pairs pair; walk back to the two k_features before it
idx = [i for i, r in enumerate(rows) if "k_features" in r["Kernel_Name"]]
i0 = idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "rocprim" in n:
        n = "rocprim::" + ("onesweep" if "onesweep" in n else "histogram" if "histogram" in n else "scan" if "scan" in n else "other")
    print("q%-3s %-28s %8.1f -> %8.1f  (%6.1f us) grid %s" % (r["Queue_Id"], n[:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
