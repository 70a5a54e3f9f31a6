#!/bin/bash
# composition of one LM iteration of the C4 window (bench.py's window section) + the kernels of its build: gpurun -- 'bash profiles/dev/c4_trace.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/kc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -o c -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > /dev/null 2>&1
f=$(find /tmp/kc -name "*kernel_trace.csv" | head -1)
python profiles/dev/lm_iteration.py $f 3
python profiles/dev/timeline.py $f 12000 > $R/gpurun_out/c4_timeline.txt
