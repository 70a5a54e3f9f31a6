#!/bin/bash
# A/B of a compile-time variant on the GPU box: bash profiles/dev/ab_build.sh "<extra hipcc flags>" <command ...>
# copies the tree to /tmp/ab, rebuilds csrc with the extra flags there and runs the command from that copy
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
FL="$1"; shift
rm -rf /tmp/ab && mkdir -p /tmp/ab && cp -r $R/wildcat-slam_amd $R/include $R/oracle $R/profiles $R/tests $R/bench.py /tmp/ab/
cd /tmp/ab/wildcat-slam_amd/csrc && rm -f match.o libwildcat_hip.so && make -j8 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $FL" libwildcat_hip.so > /tmp/ab/build.log 2>&1 || { tail -5 /tmp/ab/build.log; exit 1; }
cd /tmp/ab && "$@"
