#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the extraction on the two clouds the bench quotes
# next to C2: the 1 M-point firing-order sweep (G1) and the 10 M-point cloud (C5).  Usage: gpurun -- 'bash profiles/collect_clouds.sh'
# -> gpurun_out/clouds/*.csv; then python profiles/summarize_clouds.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clouds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cloud in "room 1000000" "g2 10000000"; do
  tag=$(echo $cloud | cut -d' ' -f1)
  for m in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $m --output-format csv -d /tmp/pc_${tag}_$m -o b -- python $R/profiles/exp_g1.py $cloud 12 > /tmp/pc.log 2>&1
    cp /tmp/pc_${tag}_$m/b_counter_collection.csv $O/${tag}_$m.csv
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_${tag}_t -o b -- python $R/profiles/exp_g1.py $cloud 12 > /tmp/pc.log 2>&1
  cp /tmp/pc_${tag}_t/b_kernel_stats.csv $O/${tag}_stats.csv
done
ls -la $O
