#!/usr/bin/env python3
"""gpurun_out/clouds/*.csv (profiles/collect_clouds.sh) -> profiles/<tag>_pmc_clouds.md   (python profiles/summarize_clouds.py r3)"""
import collections, csv, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "clouds")
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
out = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) - extraction of the 1 M-point firing-order sweep, of the 10 M-point cloud (48-byte records and 20 B / point) and of ten C2 sweeps in one launch chain (" + TAG + ")\n",
       "`python profiles/exp_g1.py <cloud> 12` (points uploaded before every sweep: cold input).  Reads x2-corrected (gfx950, MI355X_MICROARCH.md).\n"]
for tag, n_pts, title, bpp in (("room", 1_000_000, "G1 room, 1 M points in firing order", 48), ("g2", 9_999_872, "C5 cloud, 10 M points (G2), 48-byte records", 48),
                              ("g2soa", 9_999_872, "C5 cloud, 10 M points (G2), 20 B / point (float xyz + double time)", 20),
                              ("batch", 9_999_360, "ten C2 sweeps (10 x 999 936 points, 48-byte records) through one launch chain (wc_extract_surfels_batch)", 48)):
    if not os.path.exists(os.path.join(src, f"{tag}_FETCH_SIZE.csv")):
        continue
    vals = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for m in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(os.path.join(src, f"{tag}_{m}.csv"))):
            if r["Counter_Name"] == m:
                vals[short(r["Kernel_Name"])][m].append(float(r["Counter_Value"]))
    st = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(src, f"{tag}_stats.csv")))}
    sweeps = max(len(v["FETCH_SIZE"]) for k, v in vals.items() if k.startswith("k_fx") or k.startswith("k_slot"))
    out.append(f"\n## {title}\n\n| kernel | launches | avg us | read MB (x2) | written MB |\n|---|---:|---:|---:|---:|\n")
    tot = 0.0
    for k, v in sorted(vals.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"] or [0])):
        if not (k.startswith("k_fx") or k.startswith("k_slot")):
            continue
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 2 * 1024 / 1e6
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024 / 1e6
        tot += (fe + wr) * len(v["FETCH_SIZE"]) / sweeps  # (the layer-2 pair of a cloud without layer-2 nodes runs once: pro rata)
        out.append(f"| `{k}` | {len(v['FETCH_SIZE'])} | {st.get(k, 0):.1f} | {fe:.1f} | {wr:.1f} |\n")
    t_us = sum(st.get(k, 0) * len(v["FETCH_SIZE"]) / sweeps for k, v in vals.items() if k.startswith("k_fx") or k.startswith("k_slot"))
    n_surf = {"room": 15949, "g2": 312496, "g2soa": 312496, "batch": 312480}[tag]
    algo = (20 * n_pts + 144 * n_surf) / 1e6
    out.append(f"\nSum {tot:.0f} MB per call (the input alone is {bpp} B x {n_pts} = {bpp * n_pts / 1e6:.0f} MB, consumed in place); algorithmic 20 B x {n_pts} points + 144 B x {n_surf} surfels = {algo:.0f} MB: {tot / algo:.2f} x.  Kernel time {t_us:.0f} us per call -> {algo / t_us:.2f} TB/s algorithmic = {algo / t_us / 8.0:.3f} of HBM peak.\n")
open(os.path.join(root, "profiles", TAG + "_pmc_clouds.md"), "w").writelines(out)
print("".join(out))
