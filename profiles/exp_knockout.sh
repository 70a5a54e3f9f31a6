cd /tmp && export TMPDIR=/tmp
for d in ${KNOBS:-0 128 1024 1152 1408 1920}; do
WC_DEBUG_SKIP=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g$d -o b -- python $GRAFT_REPO_ROOT/profiles/exp_g1.py ${CLOUD:-room 1000000} ${STEPS:-5} ${LAYOUT:-aos} >/tmp/l2 2>&1; echo "dbg $d $(tail -1 /tmp/l2 | cut -c1-60)"; cut -d, -f1-4,6 /tmp/g$d/b_kernel_stats.csv | sed "s/(anonymous namespace):://g" | cut -c1-100 | grep k_fx
done
