export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python profiles/dev/step_var.py 15 | cut -c1-60
rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o s -- python profiles/dev/step_var.py 4 > /dev/null 2>&1
f=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); echo "== step iteration"; python profiles/dev/lm_iteration.py $f 3
rm -rf /tmp/kf; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kf -o f -- python profiles/time_facade.py > /tmp/tf.log 2>&1
grep -E "median" /tmp/tf.log | tail -2
f=$(find /tmp/kf -name "*kernel_trace.csv" | head -1); echo "== facade iteration"; python profiles/dev/lm_iteration.py $f 3; python profiles/dev/lm_iteration.py $f 12
