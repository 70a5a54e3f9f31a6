#!/bin/bash
# per-kernel register / LDS / scratch use of a HIP object: kregs.sh file.o
set -e
o=$1
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=/tmp/kregs.fat $o
t=$($B/clang-offload-bundler --list --type=o --input=/tmp/kregs.fat | grep amdgcn | head -1)
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/kregs.fat --targets=$t --output=/tmp/kregs.co
$B/llvm-readelf --notes /tmp/kregs.co | awk '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2}
/\.agpr_count:/ {a=$2}
/\.sgpr_count:/ {s=$2}
/\.group_segment_fixed_size:/ {l=$2}
/\.private_segment_fixed_size:/ {p=$2}
/\.symbol:/ {printf "%-90s vgpr %4s agpr %3s sgpr %3s lds %6s scratch %s\n", substr(name,1,90), v, a, s, l, p}'
