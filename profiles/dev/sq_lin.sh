#!/bin/bash
# SQ counters of the assembly kernels on the C4 window: bash profiles/dev/sq_lin.sh [tree]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-.}
rm -rf /tmp/sq1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/sq1 -o b -- python $R/profiles/dev/ab_lin.py $R/$T > /tmp/sq1.log 2>&1
rm -rf /tmp/sq2
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/sq2 -o b -- python $R/profiles/dev/ab_lin.py $R/$T > /tmp/sq2.log 2>&1
rm -rf /tmp/sq3
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/sq3 -o b -- python $R/profiles/dev/ab_lin.py $R/$T > /tmp/sq3.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/sq1", "/tmp/sq2", "/tmp/sq3"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv", open(d + ".log").read()[-400:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "k_lin" in k or "k_gather" in k:
            acc[k + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
