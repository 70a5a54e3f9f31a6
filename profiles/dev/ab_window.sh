#!/bin/bash
# a compile-time variant of window.hip on the GPU box: bash profiles/dev/ab_window.sh "<extra hipcc flags>" <command ...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
FL="$1"; shift
rm -rf /tmp/abw && mkdir -p /tmp/abw && cp -r $R/wildcat-slam_amd $R/include $R/oracle $R/profiles $R/tests $R/bench.py /tmp/abw/
cd /tmp/abw/wildcat-slam_amd/csrc && rm -f window.o libwildcat_hip.so && make -j8 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $FL" libwildcat_hip.so > /tmp/abw/build.log 2>&1 || { tail -5 /tmp/abw/build.log; exit 1; }
cd /tmp/abw && "$@"
