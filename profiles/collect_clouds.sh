#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the extraction on the two clouds the bench quotes
# next to C2: the 1 M-point firing-order sweep (G1) and the 10 M-point cloud (C5).  Usage: gpurun -- 'bash profiles/collect_clouds.sh'
# -> gpurun_out/clouds/*.csv; then python profiles/summarize_clouds.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clouds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (round 3: + the 10 M-point cloud as 20 B / point, and ten C2 sweeps through one launch chain)
run() {  # tag, command...
  tag=$1; shift
  for m in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc_${tag}_$m
    rocprofv3 --kernel-trace --pmc $m --output-format csv -d /tmp/pc_${tag}_$m -o b -- "$@" > /tmp/pc.log 2>&1
    cp $(find /tmp/pc_${tag}_$m -name "b_counter_collection.csv" | head -1) $O/${tag}_$m.csv
  done
  rm -rf /tmp/pc_${tag}_t
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_${tag}_t -o b -- "$@" > /tmp/pc.log 2>&1
  cp $(find /tmp/pc_${tag}_t -name "b_kernel_stats.csv" | head -1) $O/${tag}_stats.csv
}
run room python $R/profiles/exp_g1.py room 1000000 12
run g2 python $R/profiles/exp_g1.py g2 10000000 12
run g2soa python $R/profiles/exp_g1.py g2 10000000 12 soa
run batch python $R/profiles/dev/prof_batch.py 10 12
ls -la $O
