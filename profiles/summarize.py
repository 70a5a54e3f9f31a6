#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that profiles/collect.sh left under gpurun_out/<tag>/ into the committed summaries
profiles/<tag>_kernel_stats.md, profiles/<tag>_pmc.md and profiles/<tag>_bench.json."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
out = os.path.join(root, "profiles")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in name:
        for key in ("onesweep", "block_sort", "block_merge", "histogram", "scan", "lookback"):
            if key in name:
                return "rocprim::" + key + (" <u64>" if "unsigned long" in name.split("trampoline_kernel")[-1][:200] else "")
        return "rocprim::other"
    return name.split("(")[0]


rows = list(csv.DictReader(open(os.path.join(src, "trace", "b_kernel_stats.csv"))))
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
with open(os.path.join(out, f"{tag}_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats — `python bench.py --steps 100 --warmup 10 --no-cpu-baseline` ({tag})\n\n")
    f.write("| kernel | calls | avg µs | total ms | % |\n|---|---:|---:|---:|---:|\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"| `{k}` | {c} | {t / c / 1e3:.1f} | {t / 1e6:.2f} | {100 * t / tot:.1f} |\n")
shutil.copy(os.path.join(src, "trace", "b_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))

with open(os.path.join(out, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) ({tag})\n\n")
    f.write("Units: KiB per dispatch as reported.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
            "(MI355X_MICROARCH.md §HBM); the x2-corrected figure is given next to it.\n\n")
    f.write("| kernel | dispatches | FETCH_SIZE KiB | x2 MB | WRITE_SIZE KiB | MB |\n|---|---:|---:|---:|---:|---:|\n")
    vals = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        p = os.path.join(src, name, "b_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == ctr:
                vals[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
    traffic = {}
    for k, v in sorted(vals.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"] or [0])):
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        f.write(f"| `{k}` | {len(v['FETCH_SIZE'])} | {fe:.0f} | {2 * fe * 1024 / 1e6:.1f} | {wr:.0f} | {wr * 1024 / 1e6:.1f} |\n")
        traffic[k] = {"read_bytes_per_launch": round(2 * fe * 1024), "write_bytes_per_launch": round(wr * 1024),
                      "launches_sampled": len(v["FETCH_SIZE"])}
# per-kernel HBM traffic (x2-corrected reads + writes) for bench.py's roofline.traffic field
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, profiles/collect.sh {tag}; reads x2 (gfx950 correction)",
           "kernels": traffic}, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
# derived metrics (one pass each): mean per kernel
derived = collections.defaultdict(dict)
for m in ("OccupancyPercent", "MeanOccupancyPerCU", "VALUBusy", "LDSBankConflict"):
    p = os.path.join(src, "pmc_" + m, "b_counter_collection.csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == m:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        derived[k][m] = sum(v) / len(v)
if derived:
    with open(os.path.join(out, f"{tag}_occupancy.md"), "w") as f:
        f.write(f"# rocprofv3 --pmc OccupancyPercent / MeanOccupancyPerCU / VALUBusy / LDSBankConflict, one pass each ({tag})\n\n")
        f.write("Mean over the dispatches of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --no-extras` "
                "(C2 extraction + C4 window).\n\n| kernel | OccupancyPercent | MeanOccupancyPerCU | VALUBusy % | LDSBankConflict % |\n|---|---:|---:|---:|---:|\n")
        for k in sorted(derived, key=lambda k: -agg.get(k, [0, 0.0])[1]):
            d = derived[k]
            f.write("| `%s` | %s | %s | %s | %s |\n" % (k, *("%.1f" % d[m] if m in d else "-" for m in ("OccupancyPercent", "MeanOccupancyPerCU", "VALUBusy", "LDSBankConflict"))))
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    line = [l for l in open(bj) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
print("wrote", [x for x in os.listdir(out) if x.startswith(tag)])
