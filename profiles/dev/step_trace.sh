#!/bin/bash
# every kernel of the LAST odometry step of a kernel trace (start, duration, queue, grid): gpurun -- 'bash profiles/dev/step_trace.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o s -- python profiles/dev/step_var.py 3 > /dev/null 2>&1
f=$(find /tmp/ks -name "*kernel_trace.csv" | head -1)
python profiles/dev/timeline.py $f 5200 > $R/gpurun_out/step_timeline.txt
python profiles/dev/lm_iteration.py $f 3
wc -l $R/gpurun_out/step_timeline.txt
