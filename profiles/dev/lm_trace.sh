#!/bin/bash
# composition of one LM iteration under development options (round 6): gpurun -- 'bash profiles/dev/lm_trace.sh c4 "lm_back_chunks=1"'
# $1: case (c4 | small | step), $2: option spec for profiles/dev/ab_lm.py, $3: iteration from the end (default 3)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/kc; AB_LM_CASES=$1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -o c -- python profiles/dev/ab_lm.py "$2" > /dev/null 2>&1
f=$(find /tmp/kc -name "*kernel_trace.csv" | head -1)
echo "== $1 [$2]"
python profiles/dev/lm_iteration.py $f ${3:-3} $4
