#!/bin/bash
# every kernel of one facade sweep in front of its solve, with the gap in front of it: gpurun -- 'bash profiles/dev/facade_trace.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/kf; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kf -o f -- python profiles/time_facade.py > /tmp/tf.log 2>&1
grep -E "median" /tmp/tf.log | tail -1
f=$(find /tmp/kf -name "*kernel_trace.csv" | head -1)
python profiles/dev/facade_timeline.py $f 3 > $R/gpurun_out/facade_timeline.txt
python profiles/dev/lm_iteration.py $f 5
