#!/bin/bash
# k_fx_merge's threshold (records per list that make the next sweep merge first): the firing-order sweep and the facade's extraction stage
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for mm in 3 6 12 24 100000; do
  echo -n "merge_min $mm: firing order ms "
  WC_FX_MERGE_MIN=$mm python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-window 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['firing_order']['ms_per_step'], b['firing_order']['stages_ms'])"
  echo -n "   facade extract + poses "
  WC_FX_MERGE_MIN=$mm WC_ODOM_DEBUG=1 python profiles/time_facade.py 2>&1 | grep "\[odom\]" | tail -3 | sed "s/.*extract + poses \([0-9.]*\).*/\1/" | tr "\n" " "; echo
done
