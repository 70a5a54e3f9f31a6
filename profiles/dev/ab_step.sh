#!/bin/bash
# A/B of two builds on ONE box (boxes differ by more than most changes): the tree's library against the copy of an older tree under ab_old/
# (git archive <rev> + make): window tests of the new build, then step_var / time_facade alternating.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_step_gpu.py tests/test_facade_gpu.py tests/test_kat_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python profiles/dev/lm_repeat.py ${REPS:-150} 2>&1 | tail -4
for rep in 1 2; do
  for t in ab_old .; do
    [ -d $t/profiles ] || continue
    echo "== $t step"; python $t/profiles/dev/step_var.py 15 | cut -c1-44
    echo "== $t facade"; python $t/profiles/time_facade.py 2>&1 | grep -E "median" | tail -1
  done
done
