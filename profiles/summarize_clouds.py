#!/usr/bin/env python3
"""gpurun_out/clouds/*.csv (profiles/collect_clouds.sh) -> profiles/r2_pmc_clouds.md"""
import collections, csv, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "clouds")
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
out = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) - extraction of the 1 M-point firing-order sweep and of the 10 M-point cloud (r2)\n",
       "`python profiles/exp_g1.py <cloud> 12` (points uploaded before every sweep: cold input).  Reads x2-corrected (gfx950, MI355X_MICROARCH.md).\n"]
for tag, n_pts, title in (("room", 1_000_000, "G1 room, 1 M points in firing order"), ("g2", 9_999_872, "C5 cloud, 10 M points (G2)")):
    vals = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for m in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(os.path.join(src, f"{tag}_{m}.csv"))):
            if r["Counter_Name"] == m:
                vals[short(r["Kernel_Name"])][m].append(float(r["Counter_Value"]))
    st = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(src, f"{tag}_stats.csv")))}
    sweeps = max(len(v["FETCH_SIZE"]) for v in vals.values())
    out.append(f"\n## {title}\n\n| kernel | launches | avg us | read MB (x2) | written MB |\n|---|---:|---:|---:|---:|\n")
    tot = 0.0
    for k, v in sorted(vals.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"] or [0])):
        if not (k.startswith("k_fx") or k.startswith("k_slot")):
            continue
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 2 * 1024 / 1e6
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024 / 1e6
        tot += (fe + wr) * len(v["FETCH_SIZE"]) / sweeps  # (the layer-2 pair of a cloud without layer-2 nodes runs once: pro rata)
        out.append(f"| `{k}` | {len(v['FETCH_SIZE'])} | {st.get(k, 0):.1f} | {fe:.1f} | {wr:.1f} |\n")
    out.append(f"\nSum {tot:.0f} MB per sweep (the input alone is 48 B x {n_pts} = {48 * n_pts / 1e6:.0f} MB of records, consumed in place); algorithmic 20 B x {n_pts} points = {20 * n_pts / 1e6:.0f} MB (+ 144 B per surfel): {tot / (20 * n_pts / 1e6):.1f} x.\n")
open(os.path.join(root, "profiles", "r2_pmc_clouds.md"), "w").writelines(out)
print("".join(out))
