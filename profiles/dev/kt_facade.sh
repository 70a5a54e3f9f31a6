#!/bin/bash
# kernel-trace medians of chosen kernels over the facade's room stream for trees under ab_var/ (or .): bash profiles/dev/kt_facade.sh <tree> ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  n=$(echo $v | tr '/.' '__'); rm -rf /tmp/ktf_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ktf_$n -o b -- python $R/$v/profiles/time_facade.py > /tmp/ktf_$n.log 2>&1
  echo "== $v: $(grep median /tmp/ktf_$n.log | tail -1)"
  python - $n <<'PY'
import csv, glob, sys, collections, os
f = glob.glob("/tmp/ktf_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(x in k for x in os.environ.get("KT_FILTER", "k_lin,k_gather,k_pcr_init").split(",")):
        acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print("   %-40s n %5d median %.1f us mean %.1f" % (k, len(v), v[len(v) // 2], sum(v) / len(v)))
PY
done
