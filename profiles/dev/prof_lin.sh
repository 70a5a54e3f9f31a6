#!/bin/bash
# phase clocks of the assembly kernel (k_lin_fused): a throw-away -DWC_PROF_LIN build of window.hip in /tmp, the bench's window section
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/proflin && mkdir -p /tmp/proflin && cp -r $R/wildcat-slam_amd $R/include /tmp/proflin/ 2>/dev/null
mkdir -p /tmp/proflin/x && cp -r $R/wildcat-slam_amd /tmp/proflin/x/ && cp -r $R/include /tmp/proflin/x/ && cp $R/bench.py /tmp/proflin/x/ && cp -r $R/oracle /tmp/proflin/x/ && cp -r $R/profiles /tmp/proflin/x/
cd /tmp/proflin/x/wildcat-slam_amd/csrc && rm -f window.o libwildcat_hip.so && make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DWC_PROF_LIN" libwildcat_hip.so > /tmp/proflin/build.log 2>&1 || { tail -5 /tmp/proflin/build.log; exit 1; }
cd /tmp/proflin/x && python bench.py --steps 5 --no-cpu-baseline --no-extras 2>&1 | grep "^lin W" | sort | awk '{k=$2" "$6; n[k]++; for(i=7;i<=NF;i++) if ($i ~ /^[0-9]+$/) s[k,i]+=$i} END {for (k in n) print k, n[k]}' | head
python bench.py --steps 5 --no-cpu-baseline --no-extras 2>&1 | grep "^lin W" | python3 -c "
import sys, re, collections
acc = collections.defaultdict(list)
for l in sys.stdin:
    m = re.match(r'lin W=(\d+) blk \d+ count (\d+): A (\d+) \(descriptor (\d+) records (\d+) evaluate (\d+)\) sync (\d+) B (\d+) tail (\d+) total (\d+)', l)
    if m:
        w, cnt = int(m.group(1)), int(m.group(2))
        acc[(w, 'full' if cnt == 256 else ('>=128' if cnt >= 128 else '<128'))].append([int(x) for x in m.groups()[2:]])
import numpy as np
for k, v in sorted(acc.items()):
    a = np.array(v)
    print(k, len(a), 'median A %d (desc %d rec %d eval %d) sync %d B %d tail %d total %d' % tuple(np.median(a, 0)))
"
