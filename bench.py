#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X hot path (contract: see the task statement / DESIGN.md §Measurement).

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
one BuildSurfels-equivalent surfel extraction of BASELINE.json config C2 (G2 patch lattice, 3 906 root voxels x 8
patches x 32 points = 999 936 points -> 31 248 surfels) per GPU.  With N > 1 every rank extracts its own sweep
(sweeps are independent jobs in the reference, lidar_odometry.cc:523-525: a fresh GlobalMap per sweep), so the
data path has no collective and scaling is "weak".

Prints ONE JSON line (rank 0).  Extra keys: "roofline" (dominant kernel, HIP-event timed on the kernel's stream),
"cpu_baseline" (the CPU oracle timed on the same box, rank 0, N = 1 only), "stages_ms", "window" (LM-iteration
figures once the window kernels are built).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--roots", type=int, default=3906, help="root voxels per sweep (C2: 3906 -> 999 936 points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--window-scans", type=int, default=20, help="sweeps in the LM window (C4: 20)")
    ap.add_argument("--window-patches", type=int, default=50000, help="surfels per sweep (C4: 50 000 -> 1 M surfels)")
    ap.add_argument("--no-window", action="store_true", help="skip the LM-window section")
    ap.add_argument("--in-flight", type=int, default=3, help="sweeps in flight (contexts) of the extra pipelined measurement")
    args = ap.parse_args()

    import torch  # device memory + distributed plumbing only
    import torch.distributed as dist

    from wildcat_slam_amd import lib, records as R, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("WC_BENCH_FORCE_DIST") == "1"  # exercise the RCCL plumbing on one GPU (world size 1)
    if args.gpus > 1 or world > 1 or force_dist:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    # --- synthetic sweep of this rank (G2, seed + rank), resident in HBM before the timed region ------------------
    pts, info = synth.g2_lattice(args.roots, m=32, seed=synth.SEED + rank)
    n_pts = len(pts)
    exp_surfels = 8 * args.roots
    d_pts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).to(dev)
    cap = (3 * n_pts) // 20 + 1
    d_out = torch.empty(cap * 144, dtype=torch.uint8, device=dev)
    d_ids = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    ctx = lib.Context(local_rank)  # raises if libwildcat_hip.so / the GPU is missing: there is no fallback path
    base = d_pts.data_ptr()
    desc = R.Points(base, base + 24, 48, 48, n_pts)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])

    class _Ptr:
        def __init__(self, p):
            self.ptr = p

    out_p, ids_p = _Ptr(d_out.data_ptr()), _Ptr(d_ids.data_ptr())

    enq, fin = ctx.prepare_extract(desc, out_p, ids_p, cap, t_lo, t_hi)  # ctypes argument objects built once

    def step():
        enq()
        return fin()

    for _ in range(args.warmup):
        n_s = step()
    assert os.environ.get("WC_DEBUG_SKIP") or n_s == exp_surfels, (n_s, exp_surfels)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_s = step()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_pts / (elapsed / args.steps) / 1e6  # Mpts/s, whole job

    # --- two sweeps in flight (two contexts = two streams, each with its own scratch): what the chain of short, latency
    # bound kernels leaves idle is filled by the other sweep.  Reported next to the headline, never as `value`.
    pipelined = None
    try:
        if args.in_flight < 2:
            raise RuntimeError("skipped (--in-flight < 2)")
        F = args.in_flight
        ring = [(ctx, out_p, ids_p, None)]
        for _ in range(F - 1):
            c2 = lib.Context(local_rank)
            o2 = torch.empty(cap * 144, dtype=torch.uint8, device=dev)
            i2 = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
            ring.append((c2, _Ptr(o2.data_ptr()), _Ptr(i2.data_ptr()), (o2, i2)))
            for _ in range(3):
                c2.extract_enqueue(desc, ring[-1][1], ring[-1][2], cap, t_lo, t_hi)
                assert c2.extract_finish() == exp_surfels or os.environ.get("WC_DEBUG_SKIP")
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):  # sweep i is enqueued on context i mod F as soon as that context's previous sweep is done
            c2, o2, i2, _ = ring[i % F]
            if i >= F:
                c2.extract_finish()
            c2.extract_enqueue(desc, o2, i2, cap, t_lo, t_hi)
        for i in range(max(0, args.steps - F), args.steps):
            ring[i % F][0].extract_finish()
        for c2, _, _, _ in ring:
            c2.sync()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        pipelined = {"sweeps_in_flight": F, "value": round(world * n_pts / (el2 / args.steps) / 1e6, 2), "unit": "Mpts/s",
                     "ms_per_step": round(el2 / args.steps * 1e3, 5)}
        for c2, _, _, keep in ring[1:]:
            c2.close()
    except Exception as e:  # the headline must survive a failure of this extra measurement
        pipelined = {"error": repr(e)}

    # --- the same sweep in the 20 B / point layout SURVEY 8(d) counts (packed float32 xyz + float64 time instead of the
    # reference's 48-byte record): what the input layout costs.  Reported next to the headline, never as `value`.
    soa = None
    try:
        xyz = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
        d_xyz = torch.from_numpy(xyz.reshape(-1)).to(dev)
        d_t = torch.from_numpy(np.ascontiguousarray(pts["time"], np.float64)).to(dev)
        desc_soa = R.Points(d_xyz.data_ptr(), d_t.data_ptr(), 12, 8, n_pts)
        enq2, fin2 = ctx.prepare_extract(desc_soa, out_p, ids_p, cap, t_lo, t_hi)
        for _ in range(max(3, args.warmup)):
            enq2()
            n2 = fin2()
        assert os.environ.get("WC_DEBUG_SKIP") or n2 == exp_surfels, (n2, exp_surfels)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            enq2()
            fin2()
        ctx.sync()
        el3 = time.perf_counter() - t0
        soa = {"layout": "float32 xyz (stride 12) + float64 time (stride 8): 20 B / point", "value": round(n_pts / (el3 / args.steps) / 1e6, 2),
               "unit": "Mpts/s (this rank)", "ms_per_step": round(el3 / args.steps * 1e3, 5)}
    except Exception as e:
        soa = {"error": repr(e)}

    # --- per-stage device time (HIP events on the ctx stream), same steps, for the roofline object ---------------
    ctx.extract_profile(True)
    acc = {}
    k = max(10, min(args.steps, 100))
    for _ in range(k):
        step()
        for name, ms in ctx.extract_stage_ms().items():
            acc[name] = acc.get(name, 0.0) + ms
    ctx.extract_profile(False)
    stages = {name: v / k for name, v in acc.items()}
    dom = max(stages, key=stages.get)
    algo_bytes = 20 * n_pts + 144 * exp_surfels  # 20 B read per point + 144 B written per surfel
    dom_ms = stages[dom]
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    kernel_names = {"init": "k_init", "point_sort": "k_pt_runs + k_pt_bucket", "roots_stream": "k_roots<unsigned int, 1, true>",
                    "roots_emit": "k_roots_emit<unsigned int, true>", "slot_order": "k_slot_emit"}
    # measured ceiling of this device: a 1 GiB device-to-device copy, bytes read + written per second
    copy_gbs = None
    try:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * (1 << 30) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    except Exception:
        copy_gbs = None
    args._copy_gbs = copy_gbs
    roofline = {"bound": "hbm", "kernel": kernel_names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kernel_names[dom]), "algorithmic_bytes_per_launch": algo_bytes,
                "avg_kernel_ms": round(dom_ms, 5),
                "whole_pipeline_frac": round(algo_bytes / (sum(stages.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                "frac_of_measured_copy": round(achieved / copy_gbs, 5) if copy_gbs else None}

    result = {
        "metric": "surfel-extract Mpts/s",
        "value": round(value, 2),
        "unit": "Mpts/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C2: 1M-pt single scan (G2 patch lattice, %d pts -> %d surfels) per GPU, voxel-grid + 3-level octree + per-cell 3x3 PCA"
                   % (n_pts, exp_surfels), "points_per_gpu": n_pts, "surfels_per_gpu": exp_surfels, "parallelism": "sweep-per-gpu x%d" % world},
        "roofline": roofline,
        "pipelined": pipelined,
        "soa_input": soa,
        "stages_ms": {k_: round(v, 5) for k_, v in stages.items()},
    }

    # --- CPU baseline: the single-thread oracle on the same workload, rank 0, N = 1 only --------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        reps, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds:
            t1 = time.perf_counter()
            s_ref, _, _ = pyoracle.extract_surfels(pts, cap=cap)
            t_cpu += time.perf_counter() - t1
            reps += 1
        assert len(s_ref) == exp_surfels
        result["cpu_baseline"] = {"value": round(reps * n_pts / t_cpu / 1e6, 3), "unit": "Mpts/s", "cores": 1, "kind": "port",
                                  "sample": "%d full C2 sweeps (%d pts each), %.1f s of single-thread oracle (oracle/extract.cc)" % (reps, n_pts, t_cpu)}

    if not args.no_window:
        try:
            result["window"] = bench_window(ctx, args, world, rank, dev, torch, dist, cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        except Exception as e:  # the headline line must survive a failure of the extra section
            result["window"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (reads x2-corrected + writes) from the newest committed PMC summary
    (profiles/<tag>_pmc.json, written by profiles/summarize.py from separate rocprofv3 --pmc passes); None if absent."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    tot = 0
    for name in kernel.split(" + "):
        if name not in ks:
            return None
        tot += ks[name]["read_bytes_per_launch"] + ks[name]["write_bytes_per_launch"]
    return tot


def bench_window(ctx, args, world, rank, dev, torch, dist, cpu=False):
    """LM ("GN") iterations per second on a C4-like window: `scans` sweeps x `patches` surfels, binary + unary surfel
    factors + IMU factors, correspondences from the GPU matcher.  With N > 1 the correspondences are sharded over the
    ranks (unknowns replicated) and every linearisation ends in ONE RCCL all-reduce of the packed {H, g, cost}."""
    from wildcat_slam_amd import dist as wdist, records as R, synth

    t_gen = time.perf_counter()
    w = synth.surfel_window(args.window_scans, args.window_patches, seed=synth.SEED + 7, fixed_patches=args.window_patches)
    n_s = len(w["surf"])
    d_surf, d_pose = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    t_gen = time.perf_counter() - t_gen
    # correspondences (timed: part of the hot path, lidar_odometry.cc:532-538)
    d_pairs, d_pf = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    # (one untimed pass first: the library's scratch buffers are allocated on first use)
    ctx.match_device(d_surf, d_pose, n_s, d_surf, d_pose, n_s, True, d_pairs, n_s)
    ctx.match_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), False, d_pf, n_s)
    ctx.sync()
    t0 = time.perf_counter()
    n_b = ctx.match_device(d_surf, d_pose, n_s, d_surf, d_pose, n_s, True, d_pairs, n_s)
    n_u = ctx.match_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), False, d_pf, n_s)
    t_match = time.perf_counter() - t0
    # shard the correspondences (contiguous slices) and the IMU factors
    lo_b, cnt_b = wdist.shard_range(n_b, rank, world)
    lo_u, cnt_u = wdist.shard_range(n_u, rank, world)

    class _Off:
        def __init__(self, ptr):
            self.ptr = ptr

    if world > 1 or os.environ.get("WC_BENCH_FORCE_DIST") == "1":
        ctx.window_set_allreduce(wdist.make_allreduce(torch, dist, dev))
    imu_r = wdist.shard_imu(w["imu"], rank, world)  # IMU factors: a contiguous share of the state triples per rank
    build_args = (d_surf, d_pose, _Off(d_pairs.ptr + 8 * lo_b), cnt_b, imu_r if len(imu_r) >= 3 else None, w["sample_times"], w["grav"],
                  False, d_fs, d_fp, _Off(d_pf.ptr + 8 * lo_u), cnt_u)
    ctx.window_build(*build_args)  # (first call allocates)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.window_build(*build_args)  # once per solve: interval keys, sort, packed records, pieces, gather lists
    ctx.sync()
    t_build = time.perf_counter() - t0
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)
    # assembly alone: K linearisations, HIP-event timed on the ctx stream
    K = 20
    ctx.window_linearize_only(x0)
    ctx.timer_start()
    for _ in range(K):
        ctx.window_linearize_only(x0)
    lin_ms = ctx.timer_stop_ms() / K
    nb, nu, ni, npieces = ctx.window_counts()
    algo = 136 * nb + 96 * nu + 128 * ni  # bytes per binary / unary / IMU factor, per linearisation
    # full LM solve (one untimed solve first: after the CPU-baseline section the device has idled for seconds and the first
    # burst of short kernels runs at ramping clocks)
    ctx.window_solve(x0)
    ctx.sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    x, summ, _ = ctx.window_solve(x0)
    ctx.sync()
    t_solve = time.perf_counter() - t0
    iters = max(1, summ.iterations)
    out = {
        "workload": "C4-like: %d sweeps x %d surfels = %d surfels, %d sample states (%d unknowns)" % (args.window_scans, args.window_patches, n_s, ns, 12 * ns),
        "factors_total": {"binary": n_b, "unary": n_u}, "factors_this_rank": {"binary": nb, "unary": nu, "imu": ni, "pieces": npieces},
        "lm_iterations": summ.iterations, "lm_iters_per_s": round(iters / t_solve, 2), "solve_ms": round(t_solve * 1e3, 3),
        "cost": [summ.initial_cost, summ.final_cost], "termination": summ.termination,
        "linearize_ms": round(lin_ms, 4), "assembly_corr_per_s": round(world * (nb + nu) / (lin_ms * 1e-3), 1),
        "assembly_roofline": {"bound": "hbm", "achieved": round(algo / (lin_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(algo / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_linearisation": algo,
                              "frac_of_measured_copy": round(algo / (lin_ms * 1e-3) / 1e9 / args._copy_gbs, 5) if getattr(args, "_copy_gbs", None) else None},
        "build_ms": round(t_build * 1e3, 3), "match_s": round(t_match, 4), "match_surfels_per_s": round(2 * n_s / t_match, 1), "generate_s": round(t_gen, 2),
    }
    if cpu:
        # CPU baseline of the LM step: the single-thread oracle (oracle/window.cc + oracle/match.cc) on a BOUNDED sample - the
        # same 20-sweep window geometry (same 127 sample states / 1524 unknowns, same IMU factors) with 1/20 of the surfels
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        frac = 20
        ws = synth.surfel_window(args.window_scans, max(1, args.window_patches // frac), seed=synth.SEED + 7,
                                 fixed_patches=max(1, args.window_patches // frac))
        prm = pyoracle.default_params()
        t1 = time.perf_counter()
        pb = pyoracle.match(ws["surf"], ws["pose"], ws["surf"], ws["pose"], True, prm)
        pu = pyoracle.match(ws["surf"], ws["pose"], ws["fix_surf"], ws["fix_pose"], False, prm)
        t_cm = time.perf_counter() - t1
        Wc = pyoracle.Window(ws["sample_times"], ws["grav"], False, prm)
        Wc.add_binary(ws["surf"], ws["pose"], pb)
        Wc.add_unary(ws["fix_surf"], ws["fix_pose"], ws["surf"], ws["pose"], pu)
        Wc.add_imu(ws["imu"])
        t1 = time.perf_counter()
        _, sc, _ = Wc.solve(np.zeros(12 * len(ws["sample_times"])))
        t_cs = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(max(1, sc.iterations) / t_cs, 3), "unit": "LM iterations/s", "cores": 1, "kind": "port",
            "sample": "same window geometry with 1/%d of the surfels (%d surfels, %d + %d surfel factors, %d unknowns): %d LM iterations in %.1f s; "
                      "matcher %.0f surfels/s" % (frac, len(ws["surf"]), len(pb), len(pu), 12 * len(ws["sample_times"]), sc.iterations, t_cs,
                                                   2 * len(ws["surf"]) / t_cm)}
    return out


if __name__ == "__main__":
    main()
