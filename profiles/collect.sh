#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench command, then separate PMC
# passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# The traced runs use --no-extras: kernel averages are those of C2 (extraction) and C4 (window), not a mix of sizes.
# Usage: gpurun -- 'bash profiles/collect.sh r1'   -> gpurun_out/<tag>/...; then python profiles/summarize.py <tag>
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --in-flight 1 --no-extras > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --no-extras > $O/pmc_write.log 2>&1
# occupancy / VALU-busy / LDS bank conflicts of the window kernels: one derived metric per pass
for m in OccupancyPercent VALUBusy LDSBankConflict MeanOccupancyPerCU; do
rocprofv3 --kernel-trace --pmc $m --output-format csv -d $O/pmc_$m -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --no-extras > $O/pmc_$m.log 2>&1
done
rm -f $O/*/b_kernel_trace.csv.bak
ls $O $O/trace | head -20
tail -1 $O/bench.json | cut -c1-300
