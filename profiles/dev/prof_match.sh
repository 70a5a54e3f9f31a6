#!/bin/bash
# kernel trace + counters of the matcher alone (profiles/dev/time_match.py): bash profiles/dev/prof_match.sh <tag>
set -u
TAG=${1:-m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/profiles/dev/time_match.py > $O/time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python $R/profiles/dev/time_match.py > $O/trace.log 2>&1
for m in VALUBusy MemUnitBusy OccupancyPercent MemUnitStalled; do
rocprofv3 --kernel-trace --pmc $m --output-format csv -d $O/pmc_$m -o b -- python $R/profiles/dev/time_match.py > $O/pmc_$m.log 2>&1
done
cat $O/time.txt
python - <<EOP
import csv, glob, collections
f = glob.glob("$O/trace/*kernel_stats.csv") or glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:22]:
    print("%-60s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
for m in ("VALUBusy", "MemUnitBusy", "OccupancyPercent", "MemUnitStalled"):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % m, recursive=True)
    if not f:
        print(m, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
    print(m, {k: round(sum(v) / len(v), 1) for k, v in acc.items() if "knn" in k or "kd_" in k})
EOP
