#!/bin/bash
# the timelines DESIGN 3.4 ("Round 4, last part") quotes, from kernel traces of the round's final code:
# gpurun -- 'bash profiles/dev/collect_timelines.sh'  ->  gpurun_out/${TAG}_step_timeline.txt, ${TAG}_facade_timeline.txt, ${TAG}_lm_iterations.txt
export TMPDIR=/tmp
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o s -- python profiles/dev/step_var.py 3 > /dev/null 2>&1
fs=$(find /tmp/ks -name "*kernel_trace.csv" | head -1)
python profiles/dev/timeline.py $fs 5200 > $O/${TAG}_step_timeline.txt
{ echo "== odometry step (64 sample states): one LM iteration"; python profiles/dev/lm_iteration.py $fs 3; } > $O/${TAG}_lm_iterations.txt
rm -rf /tmp/kf; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kf -o f -- python profiles/time_facade.py > /tmp/tf.log 2>&1
ff=$(find /tmp/kf -name "*kernel_trace.csv" | head -1)
python profiles/dev/facade_timeline.py $ff 3 > $O/${TAG}_facade_timeline.txt
{ echo "== facade, room stream (~50 sample states): one LM iteration"; python profiles/dev/lm_iteration.py $ff 5; } >> $O/${TAG}_lm_iterations.txt
rm -rf /tmp/kc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -o c -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > /dev/null 2>&1
fc=$(find /tmp/kc -name "*kernel_trace.csv" | head -1)
{ echo "== C4 window (127 sample states): one LM iteration"; python profiles/dev/lm_iteration.py $fc 3; } >> $O/${TAG}_lm_iterations.txt
cat $O/${TAG}_lm_iterations.txt
