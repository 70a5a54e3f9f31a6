#!/bin/bash
# what a reduction level is made of: kernel time of k_pcr_level as built, without its inverses, without the right-hand sides' update
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for fl in "" "-DWC_PCR_NOINV" "-DWC_PCR_NOR" "-DWC_PCR_NOINV -DWC_PCR_NOR"; do
  bash $R/profiles/dev/ab_window.sh "$fl" bash -c 'rm -rf /tmp/kp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o p -- python profiles/dev/step_var.py 3 > /dev/null 2>&1; f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1); python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r[\"Name\"]
    if \"k_pcr\" in n: print(n.split(\"(\")[0][-22:], r[\"Calls\"], \"%.2f us\" % (float(r[\"AverageNs\"])/1e3))
" $f' | sed "s/^/[$fl] /"
done
