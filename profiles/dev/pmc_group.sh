set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/g1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export WC_KNN_GROUP=1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python $R/profiles/dev/time_match.py fixed > $O/trace.log 2>&1
for m in VALUBusy SALUBusy MemUnitBusy MeanOccupancyPerCU; do
rocprofv3 --kernel-trace --pmc $m --output-format csv -d $O/pmc_$m -o b -- python $R/profiles/dev/time_match.py fixed > $O/pmc_$m.log 2>&1
done
python $R/profiles/dev/trace_by_size.py $(find $O/trace -name "*kernel_trace.csv") knn_tree kd_ locate features onesweep gated resolve emit flags
python - <<EOP
import csv, glob, collections
for m in ("VALUBusy", "SALUBusy", "MemUnitBusy", "MeanOccupancyPerCU"):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % m, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
    print(m, {k: round(sum(v) / len(v), 1) for k, v in acc.items() if "knn" in k})
EOP
