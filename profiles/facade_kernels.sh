#!/bin/bash
# GPU time of the host facade's sweeps by kernel: gpurun -- 'bash profiles/facade_kernels.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
rm -rf /tmp/kf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -o b -- python profiles/time_facade.py > /tmp/tf.log 2>&1
grep -E "median|add_scan" /tmp/tf.log | tail -4
python - <<'PY'
import csv, glob
tot, rows = 0.0, []
for f in glob.glob("/tmp/kf/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), r["Name"][:60]))
        tot += float(r["TotalDurationNs"]) / 1e6
rows.sort(reverse=True)
print("total GPU kernel time %.1f ms" % tot)
for t, c, n in rows[:14]:
    print("%8.2f ms %6d %s" % (t, c, n))
PY
