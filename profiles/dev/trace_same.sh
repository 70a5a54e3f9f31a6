#!/bin/bash
# kernel trace of ONE same-set search of the step-like window under a development option spec: bash profiles/dev/trace_same.sh "knn_early=0" [ab_var/<tree> | .] [c4]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cat > /tmp/one_same.py <<EOP
import os, sys, time
sys.path[:0] = ["$R/${2:-.}/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
w = synth.surfel_window(*([20, 50000] if "${3:-}" == "c4" else [8, 31248]), seed=synth.SEED + 7, fixed_patches=62496)
n_s = len(w["surf"])
d_s, d_p = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
d_b = ctx.alloc(8 * n_s)
for kv in [s.split("=") for s in sys.argv[1].split(",") if s]:
    ctx.set_dev_option(kv[0], int(kv[1]))
for rep in range(4):
    ctx.sync(); t0 = time.perf_counter()
    n = ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
    print("same-set search %.3f ms, %d pairs" % ((time.perf_counter() - t0) * 1e3, n))
EOP
rm -rf /tmp/km; rocprofv3 --kernel-trace --output-format csv -d /tmp/km -o m -- python /tmp/one_same.py "$1" 2>&1 | grep "same-set"
f=$(find /tmp/km -name "*kernel_trace.csv" | head -1)
python - $f <<EOP
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_features" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?|rocprim::\w+|__amd\w+)", r["Kernel_Name"])
    print("%8.1f + %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, m.group(1) if m else r["Kernel_Name"][:40]))
EOP
