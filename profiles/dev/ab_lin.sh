#!/bin/bash
# alternates compile-time variants of the assembly (ab_var/<name>, built by mk_var.sh) on one box
export TMPDIR=/tmp
for v in ${VARS:-base r2w4 r2w3 r4w4 r2w5 base r2w4}; do timeout 300 python profiles/dev/ab_lin.py ab_var/$v 2>&1 | tail -1; done
for v in ${APART:-r2w4}; do timeout 300 python profiles/dev/ab_lin.py ab_var/$v lin_imu_apart=1 2>&1 | tail -1; done
