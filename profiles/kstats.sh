#!/bin/bash
# Quick per-kernel averages of one bench run on the GPU box: gpurun -- 'bash profiles/kstats.sh [pattern] [bench flags...]'
# (bench flags default to --no-extras; e.g. `k_ --no-window --no-clouds` profiles the odometry step)
# (--in-flight 1: the pipelined section's three contexts polling their mailboxes do not finish under the profiler)
set -u
PAT=${1:-k_}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats
[ $# -eq 0 ] && set -- --no-extras
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o b -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --in-flight 1 "$@" > /tmp/kstats.log 2>&1
python - "$PAT" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/kstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Name"]:
            print(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.2f} us')
PY
