#!/bin/bash
# builds a compile-time variant of the library HERE (hipcc cross-compiles) into ab_var/<name>/: bash profiles/dev/mk_var.sh <name> "<extra hipcc flags>" [git rev]
# (ab_var/ travels to the GPU box with the snapshot; profiles/dev/ab_*.py <tree> then run the variants alternating on one box)
set -eu
R=$(cd $(dirname $0)/../.. && pwd)
N=$1; FL=${2:-}; REV=${3:-}
rm -rf $R/ab_var/$N && mkdir -p $R/ab_var/$N
if [ -n "$REV" ]; then (cd $R && git archive $REV wildcat-slam_amd include) | tar -x -C $R/ab_var/$N
else cp -r $R/wildcat-slam_amd $R/include $R/ab_var/$N/; fi
cd $R/ab_var/$N/wildcat-slam_amd/csrc && rm -f *.o *.so
make -j16 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $FL" libwildcat_hip.so > ../../build.log 2>&1 || { tail -5 ../../build.log; exit 1; }
rm -f *.o
make -C ../host > ../../build_host.log 2>&1 || { tail -5 ../../build_host.log; exit 1; }
mkdir -p $R/ab_var/$N/profiles/dev && cp $R/profiles/time_facade.py $R/ab_var/$N/profiles/ && cp $R/profiles/dev/step_var.py $R/profiles/dev/ab_lin.py $R/ab_var/$N/profiles/dev/
python $R/profiles/dev/kregs.py window.o k_lin_fused 2>/dev/null || true
echo "built ab_var/$N ($FL)"
