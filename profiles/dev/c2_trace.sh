#!/bin/bash
# the kernels of a few consecutive C2 extractions (bench.py's timed loop) with the gaps between them: gpurun -- 'bash profiles/dev/c2_trace.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/k2; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/k2 -o c -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-window --in-flight 1 > /dev/null 2>&1
f=$(find /tmp/k2 -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fx_acc" in r["Kernel_Name"]]
i0 = idx[20]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:i0 + 16]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
    print("%-30s %8.1f -> %8.1f (%5.1f us, gap %5.1f)" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
PY
