#!/bin/bash
# counters of the walk kernel of ONE kind of search of the step-like window: bash profiles/dev/pmc_same.sh [same|fixed] "<dev options>"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
KIND=${1:-same}
cat > /tmp/one_search.py <<EOP
import os, sys, time
sys.path[:0] = ["$R/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
w = synth.surfel_window(8, 31248, seed=synth.SEED + 7, fixed_patches=62496)
n_s, n_f = len(w["surf"]), len(w["fix_surf"])
d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
d_b = ctx.alloc(8 * n_s)
for kv in [s.split("=") for s in sys.argv[2].split(",") if s]:
    ctx.set_dev_option(kv[0], int(kv[1]))
for rep in range(3):
    if sys.argv[1] == "same":
        n = ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
    else:
        n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_b, n_s)
EOP
for m in VALUBusy MemUnitBusy MemUnitStalled OccupancyPercent VALUUtilization SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $m --output-format csv -d /tmp/pm -o p -- python /tmp/one_search.py $KIND "${2:-}" > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" $m <<EOP
import csv, sys
if not sys.argv[1]:
    print(sys.argv[2], "no file"); sys.exit(0)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_knn_tree" in r["Kernel_Name"]]
print("%-18s %s" % (sys.argv[2], " ".join("%.4g" % x for x in v[-3:])))
EOP
done
