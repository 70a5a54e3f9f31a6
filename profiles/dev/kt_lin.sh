#!/bin/bash
# kernel-trace averages of the assembly kernels for variants under ab_var/: bash profiles/dev/kt_lin.sh <name> ...
cd /tmp && export TMPDIR=/tmp NOSOLVE=${NOSOLVE-1}
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  rm -rf /tmp/kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o b -- python $R/profiles/dev/ab_lin.py $R/ab_var/$v > /tmp/kt_$v.log 2>&1
  tail -1 /tmp/kt_$v.log | cut -c1-120
  python - $v <<'PY'
import csv, glob, sys, collections, os
f = glob.glob("/tmp/kt_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(f in k for f in os.environ.get("KT_FILTER", "k_lin,k_gather").split(",")):
        acc[k + " grid " + r["Grid_Size_X"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print("   %-40s n %4d median %.1f us min %.1f" % (k, len(v), v[len(v) // 2], v[0]))
PY
done
