#!/bin/bash
# alternates the tree's library with ab_old/'s on one box
for rep in 1 2; do for t in ab_old .; do python profiles/dev/ab_extract.py $t ${STEPS:-200} 2>&1 | tail -1; done; done
