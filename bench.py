#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X hot path (contract: the task statement; DESIGN.md §5 Measurement).

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM: one
BuildSurfels-equivalent surfel extraction of BASELINE.json config C2 (G2 patch lattice, 3 906 root voxels x 8 patches x
32 points = 999 936 points -> 31 248 surfels) per GPU.  With N > 1 every rank extracts its own sweep (sweeps are
independent jobs in the reference, lidar_odometry.cc:523-525: a fresh GlobalMap per sweep), so the headline has no
data-path collective and scaling is "weak".

Prints ONE JSON line (rank 0).  Objects next to the contract keys:
  roofline        SURVEY §8(d): algorithmic bytes of the extraction STAGE / device time of ALL kernels of the stage (HIP events on
                  the ctx stream); the dominant kernel's own figure is a sub-object; `traffic` names the committed PMC summary it is
                  read from (it is not measured in this run)
  cpu_baseline    the CPU oracle on the same sweeps, one pinned core, rank 0 / N = 1 only
  firing_order    the same extraction on a 1 M-point sweep in the order a spinning multi-beam lidar produces (G1), so that the
                  run-structured headline is never quoted alone
  cloud_10m       BASELINE config 5's cloud (10 M points) on ONE GPU; with N > 1 `sharded_cloud` routes it over the ranks
                  (one all-to-all over RCCL, csrc/route.hip) - strong scaling of one cloud
  window          BASELINE config 4 (1 M-surfel window + IMU): matcher, one linearisation (assembly roofline), LM iterations/s
                  with their own roofline object; correspondences sharded over the ranks with ONE all-reduce per linearisation
  odometry_step   north_star's headline workload: one full odometry step (lidar_odometry.cc:523-566) on a 10 x C2 window -
                  extraction of the newest 1 M-point sweep -> pose update -> 2 x match -> build -> solve -> pose update - with its
                  stage split, roofline fraction and the CPU oracle beside it
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
# the five event-bracketed stages of wc_extract_stage_ms and the kernels of the DEFAULT (integer-moment) path inside them
STAGE_KERNELS = {"init": "(none: the control block is cleared by the previous sweep's k_slot_emit)", "point_sort": "k_fx_acc<1>",
                 "roots_stream": "k_fx_nodes<1>", "roots_emit": "k_fx_acc<2> + k_fx_nodes<2>", "slot_order": "k_slot_emit"}


class _Ptr:
    def __init__(self, p):
        self.ptr = p


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cores": os.cpu_count()}


class pinned_core:
    """the CPU-baseline legs run on ONE pinned core (the reference is single-threaded)"""

    def __enter__(self):
        self.prev = None
        try:
            self.prev = os.sched_getaffinity(0)
            self.core = max(self.prev)  # away from core 0 (interrupts) when there is a choice
            os.sched_setaffinity(0, {self.core})
        except (AttributeError, OSError):
            self.core = None
        return self

    def __exit__(self, *a):
        if self.prev is not None:
            try:
                os.sched_setaffinity(0, self.prev)
            except OSError:
                pass


def _short_kernel(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc_traffic_measured(kernels, roots):
    """HBM bytes per sweep of the extraction stage MEASURED IN THIS RUN: two child processes under `rocprofv3 --kernel-trace --pmc
    FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the TCC counters do not fit one, MI355X_MICROARCH.md) run 20 extractions of the
    same C2 sweep (profiles/exp_g1.py); per kernel the mean over its dispatches, reads x2 (the guide's gfx950 correction for wide
    coalesced reads), writes as reported.  None when rocprofv3 is not on the box or a pass fails (the caller then quotes the
    committed summary, labelled)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or os.environ.get("WC_BENCH_NO_PMC"):
        return None
    per = {}
    tmp = tempfile.mkdtemp(prefix="wc_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable,
                   os.path.join(ROOT, "profiles", "exp_g1.py"), "g2", str(roots * 256), "20"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
            found = None
            for base, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        found = os.path.join(base, f)
            if r.returncode != 0 or not found:
                return None
            acc = {}
            for row in csv.DictReader(open(found)):
                if row.get("Counter_Name") == ctr:
                    acc.setdefault(_short_kernel(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                per.setdefault(k, {})[ctr] = (sum(v) / len(v), len(v))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    tot, detail, missing = 0.0, {}, []
    most = max([per[k]["FETCH_SIZE"][1] for k in kernels if k in per and "FETCH_SIZE" in per[k]] or [1])
    for k in kernels:
        if k not in per or "FETCH_SIZE" not in per[k] or "WRITE_SIZE" not in per[k]:
            missing.append(k)
            continue
        rd, wr = 2 * per[k]["FETCH_SIZE"][0] * 1024, per[k]["WRITE_SIZE"][0] * 1024  # KiB per dispatch as reported
        share = per[k]["FETCH_SIZE"][1] / most  # a kernel that only ran in some sweeps (the layer-2 pair) counts pro rata
        detail[k] = {"read_bytes": round(rd), "write_bytes": round(wr), "dispatches": per[k]["FETCH_SIZE"][1]}
        tot += (rd + wr) * share
    if not detail:
        return None
    return {"bytes_per_launch": round(tot), "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 20 "
            "sweeps of the same C2 workload in a child process (profiles/exp_g1.py); reads x2 (gfx950 correction), per sweep",
            "per_kernel": detail, "kernels_missing_from_summary": missing}


def pmc_traffic(kernels):
    """HBM bytes per launch (reads x2-corrected + writes) summed over `kernels`, from the newest committed PMC summary
    (profiles/<tag>_pmc.json, written by profiles/summarize.py from separate rocprofv3 --pmc passes).  The fallback when the run cannot
    measure it itself (pmc_traffic_measured): the returned object names its source."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    tot, missing = 0, []
    most = max([ks[n]["launches_sampled"] for n in kernels if n in ks] or [1])
    for name in kernels:
        if name in ks:  # per SWEEP: a kernel of the stage that only ran in a few sweeps of the pass (the layer-2 pair) counts pro rata
            tot += (ks[name]["read_bytes_per_launch"] + ks[name]["write_bytes_per_launch"]) * ks[name]["launches_sampled"] / most
        else:
            missing.append(name)
    return {"bytes_per_launch": round(tot), "source": "NOT measured in this run (rocprofv3 unavailable or a pass failed): profiles/" + os.path.basename(files[-1]),
            "kernels_missing_from_summary": missing}


def extraction_stage(ctx, step, n_pts, n_surfels, steps, measure_roots=None):
    """per-stage device time (HIP events on the ctx stream around every kernel group of the stage) -> (stages_ms, roofline)"""
    ctx.extract_profile(True)
    acc = {}
    k = max(10, min(steps, 100))
    for _ in range(k):
        step()
        for name, ms in ctx.extract_stage_ms().items():
            acc[name] = acc.get(name, 0.0) + ms
    # the stage as a whole: ONE pair of events around all its kernels (an event between two kernels costs ~5 us of stream time;
    # the per-group split above carries three of them)
    ctx.extract_profile(2)
    whole = 0.0
    for _ in range(k):
        step()
        whole += ctx.extract_stage_ms()["point_sort"]
    ctx.extract_profile(False)
    stages = {name: v / k for name, v in acc.items()}
    algo = 20 * n_pts + 144 * n_surfels  # 20 B read per point + 144 B written per surfel
    stage_ms = whole / k
    dom = max((s for s in stages if s != "init"), key=stages.get)
    ach = algo / (stage_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "definition": "SURVEY 8(d): (20 B x points + 144 B x surfels) / device time of ALL kernels of the stage (one pair of HIP events "
                          "on the ctx stream around them; stages_ms is a separate run with an event after every kernel group)",
            "algorithmic_bytes_per_step": algo, "stage_device_ms": round(stage_ms, 5),
            "traffic": None,
            "dominant_kernel": {"kernel": STAGE_KERNELS.get(dom, dom), "avg_ms": round(stages[dom], 5),
                                "frac_if_it_ran_alone": round(algo / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
    # HBM traffic of the stage: measured in this run for the headline workload (measure_roots), the committed summary otherwise
    kernels = [n for s in stages if s != "init" for n in STAGE_KERNELS.get(s, s).split(" + ")]
    if measure_roots is not None:  # (-1: the headline workload on a rank that does not profile - the committed summary)
        roof["traffic"] = (pmc_traffic_measured(kernels, measure_roots) if measure_roots > 0 else None) or pmc_traffic(kernels)
        if roof["traffic"]:
            roof["traffic"]["over_algorithmic"] = round(roof["traffic"]["bytes_per_launch"] / algo, 3)
    else:
        roof["traffic"] = {"bytes_per_launch": None, "source": "not collected for this workload in the run: profiles/r6_pmc_clouds.md (firing order, 10 M points, batch)"}
    # the same facts as FLAT scalars (nested objects did not survive into the driver's `parsed` line)
    tr = roof["traffic"] or {}
    roof["traffic_bytes_per_launch"] = tr.get("bytes_per_launch")
    roof["traffic_over_algorithmic"] = tr.get("over_algorithmic")
    src = tr.get("source") or ""
    roof["traffic_source"] = ("measured in run" if src.startswith("measured in this run") else
                              "committed file" if src.startswith("NOT measured") else "not collected")
    roof["dominant_kernel_name"] = roof["dominant_kernel"]["kernel"]
    roof["dominant_kernel_avg_ms"] = roof["dominant_kernel"]["avg_ms"]
    roof["dominant_kernel_frac"] = roof["dominant_kernel"]["frac_if_it_ran_alone"]
    return stages, roof


def time_extract(ctx, desc, out_p, ids_p, cap, t_lo, t_hi, steps, warmup, expect=None):
    enq, fin = ctx.prepare_extract(desc, out_p, ids_p, cap, t_lo, t_hi)
    n_s = 0
    for _ in range(max(1, warmup)):
        enq()
        n_s = fin()
    if expect is not None and not os.environ.get("WC_BENCH_NO_ASSERT"):
        assert n_s == expect, (n_s, expect)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        enq()
        n_s = fin()
    ctx.sync()
    return (time.perf_counter() - t0) / steps, n_s, (enq, fin)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--roots", type=int, default=3906, help="root voxels per sweep (C2: 3906 -> 999 936 points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--window-scans", type=int, default=20, help="sweeps in the LM window (C4: 20)")
    ap.add_argument("--window-patches", type=int, default=50000, help="surfels per sweep (C4: 50 000 -> 1 M surfels)")
    ap.add_argument("--no-window", action="store_true", help="skip the LM-window section")
    ap.add_argument("--no-extras", action="store_true", help="skip firing_order / cloud_10m / odometry_step")
    ap.add_argument("--no-clouds", action="store_true", help="skip firing_order / cloud_10m only (profiling the odometry step)")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="N > 1: seconds the sections after the headline may take")
    ap.add_argument("--in-flight", type=int, default=3, help="sweeps in flight (contexts) of the extra pipelined measurement")
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("WC_BENCH_FORCE_DIST") == "1") and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: become the launcher - one rank per GPU through torch.distributed.run (rendezvous on
        # 127.0.0.1: the container's hostname may not resolve); rank 0 of the children prints the JSON line
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execvp(cmd[0], cmd)

    # stdout carries exactly ONE line, the JSON result: everything libraries print (RCCL greets with a version banner on
    # stdout when its first communicator comes up) goes to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch  # device memory + distributed plumbing only
    import torch.distributed as dist

    from wildcat_slam_amd import lib, records as R, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("WC_BENCH_FORCE_DIST") == "1"  # exercise the RCCL plumbing on one GPU (world size 1)
    if args.gpus > 1 or world > 1 or force_dist:
        assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    use_dist = dist.is_initialized()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def to_dev(arr):
        return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(dev)

    # --- synthetic sweep of this rank (G2, seed + rank), resident in HBM before the timed region ------------------
    pts, _ = synth.g2_lattice(args.roots, m=32, seed=synth.SEED + rank)
    n_pts = len(pts)
    exp_surfels = 8 * args.roots
    d_pts = to_dev(pts)
    cap = (3 * n_pts) // 20 + 1
    d_out = torch.empty(cap * 144, dtype=torch.uint8, device=dev)
    d_ids = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    ctx = lib.Context(local_rank)  # raises if libwildcat_hip.so / the GPU is missing: there is no fallback path
    comm_kind = None
    if use_dist:
        # the library's own RCCL communicator (csrc/comm.hip): collectives on the ctx stream, no host synchronisation; rank 0's
        # unique id travels through torch.distributed, which is plumbing here
        # (every step of the choice is agreed on by all ranks: a rank that fell back alone would leave the others in a collective)
        ids, err = [None], None
        if rank == 0:
            try:
                ids = [lib.rccl_unique_id()]
            except Exception as e:  # librccl.so not loadable
                err = e
        dist.broadcast_object_list(ids, src=0)
        ok = 0
        if ids[0] is not None:
            try:
                ctx.comm_rccl_init(rank, world, ids[0])
                ok = 1
            except Exception as e:
                err = e
        agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            comm_kind = "in-library RCCL (ncclAllReduce / grouped ncclSend+ncclRecv on the ctx stream)"
        else:
            from wildcat_slam_amd import dist as wdist

            ctx.set_comm(wdist.TorchComm(torch, dist, dev))
            comm_kind = "torch.distributed callbacks (in-library RCCL unavailable on some rank: %r)" % (err,)
    base = d_pts.data_ptr()
    desc = R.Points(base, base + 24, 48, 48, n_pts)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    out_p, ids_p = _Ptr(d_out.data_ptr()), _Ptr(d_ids.data_ptr())

    enq, fin = ctx.prepare_extract(desc, out_p, ids_p, cap, t_lo, t_hi)  # ctypes argument objects built once

    def step():
        enq()
        return fin()

    for _ in range(args.warmup):
        n_s = step()
    assert os.environ.get("WC_BENCH_NO_ASSERT") or n_s == exp_surfels, (n_s, exp_surfels)

    ctx.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_s = step()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_pts / (elapsed / args.steps) / 1e6  # Mpts/s, whole job

    stages, roofline = extraction_stage(ctx, step, n_pts, exp_surfels, args.steps, measure_roots=args.roots if (rank == 0 and world == 1) else -1)
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
    roofline["measured_copy_GBs"] = round(copy_gbs, 1) if copy_gbs else None
    roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 5) if copy_gbs else None
    roofline["driver_clock_frac"] = round((20 * n_pts + 144 * exp_surfels) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)  # incl. host turn-around

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
        "config": {"workload": "C2: 1M-pt single scan (G2 patch lattice, %d pts -> %d surfels) per GPU, voxel-grid + 3-level octree + per-cell 3x3 PCA; "
                               "input = the reference's 48-byte hilti_ros::Point records, consumed in place" % (n_pts, exp_surfels),
                   "points_per_gpu": n_pts, "surfels_per_gpu": exp_surfels, "parallelism": "sweep-per-gpu x%d" % world},
        "roofline": roofline,
        "stages_ms": {k_: round(v, 5) for k_, v in stages.items()},
        "stage_kernels": STAGE_KERNELS,
        "host": cpu_info(),
        "communicator": comm_kind,
    }

    # --- several sweeps in flight (contexts = streams, each with its own scratch).  Never `value`. --------------------------
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
                assert c2.extract_finish() == exp_surfels or os.environ.get("WC_BENCH_NO_ASSERT")
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
        el2 = max_over_ranks(time.perf_counter() - t0)
        result["pipelined"] = {"sweeps_in_flight": F, "value": round(world * n_pts / (el2 / args.steps) / 1e6, 2), "unit": "Mpts/s",
                               "ms_per_step": round(el2 / args.steps * 1e3, 5)}
        for c2, _, _, keep in ring[1:]:
            c2.close()
    except Exception as e:  # the headline must survive a failure of this extra measurement
        result["pipelined"] = {"error": repr(e)}

    # --- the same sweep in the 20 B / point layout SURVEY 8(d) counts.  Never `value`. ---------------------------------------
    try:
        xyz = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
        d_xyz = torch.from_numpy(xyz.reshape(-1)).to(dev)
        d_t = torch.from_numpy(np.ascontiguousarray(pts["time"], np.float64)).to(dev)
        desc_soa = R.Points(d_xyz.data_ptr(), d_t.data_ptr(), 12, 8, n_pts)
        sec, _, (enq2, fin2) = time_extract(ctx, desc_soa, out_p, ids_p, cap, t_lo, t_hi, args.steps, max(3, args.warmup), exp_surfels)

        def step2():
            enq2()
            return fin2()

        st2, roof2 = extraction_stage(ctx, step2, n_pts, exp_surfels, args.steps)
        result["soa_input"] = {"layout": "float32 xyz (stride 12) + float64 time (stride 8): 20 B / point", "value": round(n_pts / sec / 1e6, 2),
                               "unit": "Mpts/s (this rank)", "ms_per_step": round(sec * 1e3, 5), "stage_device_ms": roof2["stage_device_ms"],
                               "roofline_frac": roof2["frac"]}
        del d_xyz, d_t
    except Exception as e:
        result["soa_input"] = {"error": repr(e)}

    # --- the same sweep with exact_sums = 1: fp64 sums in the reference's order, the arithmetic whose STEP meets north_star's 1e-6
    # on pose increments (tests/test_step_gpu.py; the default integer-moment path is held to 1e-5 there).  Never `value`. ---------
    try:
        ctx.set_exact_sums(True)
        sec_x, _, (enqx, finx) = time_extract(ctx, desc, out_p, ids_p, cap, t_lo, t_hi, args.steps, max(3, args.warmup), exp_surfels)

        def stepx():
            enqx()
            return finx()

        _, roofx = extraction_stage(ctx, stepx, n_pts, exp_surfels, args.steps)
        result["exact_sums"] = {"what": "the headline sweep with wc_params.exact_sums = 1 (sums in the reference's order; surfels 1e-6, step corrections 1e-6 "
                                        "in tests/test_step_gpu.py) beside the default path (integer moments; surfels 1e-6, step corrections tested to 1e-5)",
                                "value": round(n_pts / sec_x / 1e6, 2), "unit": "Mpts/s (this rank)", "ms_per_step": round(sec_x * 1e3, 5),
                                "stage_device_ms": roofx["stage_device_ms"], "roofline_frac": roofx["frac"],
                                "default_path_ms_per_step": round(ms_per_step, 5), "tolerance_tested": {"exact_sums": 1e-6, "default": 1e-5}}
    except Exception as e:
        result["exact_sums"] = {"error": repr(e)}
    finally:
        ctx.set_exact_sums(False)

    cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if cpu:  # CPU baseline: the single-thread oracle on the same workload, one pinned core
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        with pinned_core() as pc:
            reps, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds:
                t1 = time.perf_counter()
                s_ref, _, _ = pyoracle.extract_surfels(pts, cap=cap)
                t_cpu += time.perf_counter() - t1
                reps += 1
        assert len(s_ref) == exp_surfels
        result["cpu_baseline"] = {"value": round(reps * n_pts / t_cpu / 1e6, 3), "unit": "Mpts/s", "cores": 1, "kind": "port", "pinned_core": pc.core,
                                  "cpu": result["host"]["model"],
                                  "sample": "%d full C2 sweeps (%d pts each), %.1f s of single-thread oracle (oracle/extract.cc)" % (reps, n_pts, t_cpu)}

    def extras():
        torch.cuda.set_device(dev)  # (N > 1: this runs in a worker thread, and the current device is per thread)
        if not args.no_extras and not args.no_clouds:
            for name, fn in (("firing_order", bench_firing_order), ("cloud_10m", bench_cloud_10m), ("batched_10x_c2", bench_batched)):
                try:
                    result[name] = fn(ctx, args, world, rank, dev, torch, dist, to_dev)
                except Exception as e:
                    result[name] = {"error": repr(e)}
        if not args.no_window:
            try:
                result["window"] = bench_window(ctx, args, world, rank, dev, torch, dist, cpu=cpu)
            except Exception as e:  # the headline line must survive a failure of the extra section
                result["window"] = {"error": repr(e)}
            w_ = result["window"]
            if "lm_iters_per_s" in w_:  # BASELINE.json names "GN iters/sec on 1M-surfel window" first: beside the extraction value
                result["lm_iters_per_s"] = w_["lm_iters_per_s"]
                result["lm_roofline"] = w_["lm_roofline"]
                result["assembly_roofline"] = w_["assembly_roofline"]
        if not args.no_extras:
            try:
                result["odometry_step"] = bench_odometry_step(ctx, args, world, rank, dev, torch, dist, cpu=cpu)
            except Exception as e:
                result["odometry_step"] = {"error": repr(e)}
            if world == 1:  # (the facade is one object on one GPU, as the reference's node holds it)
                try:
                    result["facade_stream"] = bench_facade_stream(local_rank, cpu)
                except Exception as e:
                    result["facade_stream"] = {"error": repr(e)}
                try:
                    result["multi_gpu_model"] = multi_gpu_model(result, local_rank)
                except Exception as e:
                    result["multi_gpu_model"] = {"error": repr(e)}

    if world == 1:
        extras()
    else:
        # Several ranks: the extra sections use collectives (routed extraction, sharded window).  A collective that never returns
        # must not cost the headline line: the sections run in a worker thread, and a rank that waits longer than --extras-timeout
        # prints what it has and leaves.
        import threading

        th = threading.Thread(target=extras, daemon=True)
        th.start()
        th.join(args.extras_timeout)
        if th.is_alive():
            result["extras_timeout_s"] = args.extras_timeout
            if rank == 0:
                real_stdout.write(json.dumps(result, default=repr) + "\n")
                real_stdout.flush()
            os._exit(0)

    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def bench_firing_order(ctx, args, world, rank, dev, torch, dist, to_dev):
    """a 1 M-point sweep in FIRING ORDER (G1 room ray-cast, ring = i mod 32): what a spinning multi-beam lidar delivers.  One run
    per point, heavy-tailed voxel occupancy, octree layer 2 in use."""
    from wildcat_slam_amd import records as R, synth

    pts = synth.g1_room(1_000_000, seed=synth.SEED + 3)
    n = len(pts)
    d = to_dev(pts)
    cap = (3 * n) // 20 + 1
    d_out, d_ids = torch.empty(cap * 144, dtype=torch.uint8, device=dev), torch.empty(cap * 16, dtype=torch.uint8, device=dev)
    desc = R.Points(d.data_ptr(), d.data_ptr() + 24, 48, 48, n)
    steps = max(10, args.steps // 4)
    sec, n_s, (enq, fin) = time_extract(ctx, desc, _Ptr(d_out.data_ptr()), _Ptr(d_ids.data_ptr()), cap, float(pts["time"][0]), float(pts["time"][-1]), steps, 20)

    def step():
        enq()
        return fin()

    stages, roof = extraction_stage(ctx, step, n, n_s, steps)
    return {"workload": "G1 room ray-cast in firing order: %d points -> %d surfels" % (n, n_s), "ms_per_step": round(sec * 1e3, 5),
            "value": round(n / sec / 1e6, 2), "unit": "Mpts/s (this rank)", "stage_device_ms": roof["stage_device_ms"], "roofline_frac": roof["frac"],
            "stages_ms": {k: round(v, 5) for k, v in stages.items()}}


def bench_batched(ctx, args, world, rank, dev, torch, dist, to_dev):
    """K = 10 C2 sweeps through ONE launch chain (wc_extract_surfels_batch_*: C3 / C4's windows hold 5 / 20 sweeps; a single
    1 M-point sweep is launch-latency bound), as 48-byte records and as the 20 B / point layout; and the sweep-preparation chain
    of the facade - wc_undistort_sweep_packed (lidar_odometry.cc:143-158) -> extraction of its 20 B / point output - on one sweep."""
    from wildcat_slam_amd import records as R, synth

    K = 10
    sweeps = [synth.g2_lattice(args.roots, m=32, seed=synth.SEED + 300 + k)[0] for k in range(K)]
    n = len(sweeps[0])
    cap = (3 * n) // 20 + 1
    exp = 8 * args.roots
    keep, out = [], {}
    for layout in ("aos", "soa"):
        jobs = []
        for p in sweeps:
            d_o, d_i = torch.empty(cap * 144, dtype=torch.uint8, device=dev), torch.empty(cap * 16, dtype=torch.uint8, device=dev)
            if layout == "aos":
                d = to_dev(p)
                desc = R.Points(d.data_ptr(), d.data_ptr() + 24, 48, 48, n)
                keep.append((d, d_o, d_i))
            else:
                d_x = torch.from_numpy(np.stack([p["x"], p["y"], p["z"]], axis=1).astype(np.float32).reshape(-1)).to(dev)
                d_t = torch.from_numpy(np.ascontiguousarray(p["time"], np.float64)).to(dev)
                desc = R.Points(d_x.data_ptr(), d_t.data_ptr(), 12, 8, n)
                keep.append((d_x, d_t, d_o, d_i))
            jobs.append((desc, _Ptr(d_o.data_ptr()), _Ptr(d_i.data_ptr()), cap, float(p["time"][0]), float(p["time"][-1])))
        enq, fin = ctx.extract_batch_prepare(jobs)
        for _ in range(5):
            enq()
            counts = fin()
        assert all(c == exp for c in counts) or os.environ.get("WC_BENCH_NO_ASSERT"), counts
        reps = max(5, args.steps // 10)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            enq()
            fin()
        ctx.sync()
        sec = (time.perf_counter() - t0) / reps
        dev_ms = 0.0
        for _ in range(reps):  # device time of the chain: HIP events on the ctx stream around the batch's launches
            ctx.timer_start()
            enq()
            dev_ms += ctx.timer_stop_ms()
            fin()
        dev_ms /= reps
        algo = K * (20 * n + 144 * exp)
        out[layout] = {"ms_per_batch": round(sec * 1e3, 5), "ms_per_sweep": round(sec * 1e3 / K, 5), "value": round(K * n / sec / 1e6, 2), "unit": "Mpts/s (this rank)",
                       "stage_device_ms_per_batch": round(dev_ms, 5), "roofline_frac": round(algo / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                       "roofline_frac_wall_clock": round(algo / sec / 1e9 / HBM_PEAK_GBS, 5)}
    out["workload"] = "%d C2 sweeps (%d points -> %d surfels each) enqueued together: one launch chain for all of them" % (K, n, exp)
    # undistortion + extraction of one sweep, the facade's chain (48 B read + 20 B written, then 20 B read)
    try:
        p = sweeps[0]
        imu, _ = synth.imu_states(float(p["time"][0]) - 0.0031, float(p["time"][-1]) + 0.01, t_origin=float(p["time"][0]))
        d_raw, d_imu = ctx.to_device(p), ctx.to_device(imu)
        d_xyz, d_t = ctx.alloc(12 * n), ctx.alloc(8 * n)
        d_o, d_i = ctx.alloc(144 * cap), ctx.alloc(16 * cap)
        import ctypes as C

        def chain():
            ctx._ck(ctx.lib.wc_undistort_sweep_packed(ctx.h, C.c_void_p(d_raw.ptr), C.c_uint64(n), C.c_void_p(d_imu.ptr), C.c_uint64(len(imu)),
                                                      C.c_void_p(d_xyz.ptr), C.c_void_p(d_t.ptr)))
            ctx.extract_enqueue(R.Points(d_xyz.ptr, d_t.ptr, 12, 8, n), d_o, d_i, cap, float(p["time"][0]), float(p["time"][-1]))
            return ctx.extract_finish()

        for _ in range(5):
            m = chain()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            chain()
        ctx.sync()
        sec = (time.perf_counter() - t0) / 20
        out["undistort_then_extract"] = {"ms_per_sweep": round(sec * 1e3, 5), "surfels": m, "value": round(n / sec / 1e6, 2), "unit": "Mpts/s",
                                         "bytes_per_point": "48 read + 20 written (wc_undistort_sweep_packed), 20 read (extraction); round 2: 48 + 48 + 48"}
    except Exception as e:
        out["undistort_then_extract"] = {"error": repr(e)}
    return out


def bench_cloud_10m(ctx, args, world, rank, dev, torch, dist, to_dev):
    """BASELINE config 5's cloud: G2, 39 062 roots = 9 999 872 points.  N = 1: the whole cloud on this GPU.  N > 1: every rank
    holds a time-contiguous 1/N slice; wc_extract_surfels_sharded routes the points to the owner of their root voxel (ONE
    all-to-all of 24-byte records over RCCL) and each rank extracts its voxels - strong scaling of one cloud."""
    from wildcat_slam_amd import dist as wdist, records as R, synth

    n_roots = 39_062
    pts, _ = synth.g2_lattice(n_roots, m=32, seed=synth.SEED + 50)
    n = len(pts)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    steps = max(5, args.steps // 10)
    if world == 1:
        d = to_dev(pts)
        cap = (3 * n) // 20 + 1
        d_out, d_ids = torch.empty(cap * 144, dtype=torch.uint8, device=dev), torch.empty(cap * 16, dtype=torch.uint8, device=dev)
        desc = R.Points(d.data_ptr(), d.data_ptr() + 24, 48, 48, n)
        sec, n_s, (enq, fin) = time_extract(ctx, desc, _Ptr(d_out.data_ptr()), _Ptr(d_ids.data_ptr()), cap, t_lo, t_hi, steps, 5, 8 * n_roots)

        def step():
            enq()
            return fin()

        stages, roof = extraction_stage(ctx, step, n, n_s, steps)
        out = {"workload": "C5 cloud on one GPU: %d points -> %d surfels; input = 48-byte hilti_ros::Point records" % (n, n_s), "ms_per_step": round(sec * 1e3, 5),
               "value": round(n / sec / 1e6, 2), "unit": "Mpts/s", "stage_device_ms": roof["stage_device_ms"], "roofline_frac": roof["frac"],
               "dominant_kernel": roof["dominant_kernel"], "stages_ms": {k: round(v, 5) for k, v in stages.items()},
               "node_stage": "two kernels (k_fx_walk + k_fx_test) above 2 M points"}
        # the same cloud as the 20 bytes per point the extraction reads (what wc_undistort_sweep_packed leaves): never `value`
        try:
            d_xyz = torch.from_numpy(np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32).reshape(-1)).to(dev)
            d_t = torch.from_numpy(np.ascontiguousarray(pts["time"], np.float64)).to(dev)
            desc2 = R.Points(d_xyz.data_ptr(), d_t.data_ptr(), 12, 8, n)
            sec2, n_s2, (enq2, fin2) = time_extract(ctx, desc2, _Ptr(d_out.data_ptr()), _Ptr(d_ids.data_ptr()), cap, t_lo, t_hi, steps, 3, 8 * n_roots)

            def step2():
                enq2()
                return fin2()

            st2, roof2 = extraction_stage(ctx, step2, n, n_s2, steps)
            out["soa_input"] = {"layout": "float32 xyz (stride 12) + float64 time (stride 8): 20 B / point", "ms_per_step": round(sec2 * 1e3, 5),
                                "value": round(n / sec2 / 1e6, 2), "unit": "Mpts/s", "stage_device_ms": roof2["stage_device_ms"], "roofline_frac": roof2["frac"],
                                "stages_ms": {k: round(v, 5) for k, v in st2.items()}}
        except Exception as e:
            out["soa_input"] = {"error": repr(e)}
        return out
    lo, cnt = wdist.shard_range(n, rank, world)
    d_slice = ctx.to_device(pts[lo: lo + cnt])
    cap = (3 * n) // 20 // world * 2 + 4096
    bufs = (ctx.alloc(cap * 144), ctx.alloc(cap * 16), cap)
    out = None
    for _ in range(2):
        out = ctx.extract_surfels_sharded(d_slice, cnt, t_lo, t_hi, out=bufs)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = ctx.extract_surfels_sharded(d_slice, cnt, t_lo, t_hi, out=bufs)
    ctx.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tt = torch.tensor([el, float(out[2]), float(out[3])], dtype=torch.float64, device=dev)
    mx = tt.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
    sec = float(mx[0].item()) / steps
    return {"workload": "C5 cloud sharded by root voxel over %d GPUs (route: one all-to-all of 24 B / point)" % world, "ms_per_step": round(sec * 1e3, 5),
            "value": round(n / sec / 1e6, 2), "unit": "Mpts/s (whole cloud)", "surfels_total": int(tt[1].item()), "points_routed_total": int(tt[2].item()),
            "max_points_on_a_rank": int(mx[2].item()), "scaling": "strong", "roofline_frac": round((20 * n + 144 * 8 * n_roots) / sec / 1e9 / (world * HBM_PEAK_GBS), 5)}


def bench_window(ctx, args, world, rank, dev, torch, dist, cpu=False):
    """LM ("GN") iterations per second on a C4-like window: `scans` sweeps x `patches` surfels, binary + unary surfel
    factors + IMU factors, correspondences from the GPU matcher.  With N > 1 the correspondences are sharded over the
    ranks (unknowns replicated) and every linearisation ends in ONE RCCL all-reduce of the packed {H, g, cost}."""
    from wildcat_slam_amd import dist as wdist, synth

    t_gen = time.perf_counter()
    w = synth.surfel_window(args.window_scans, args.window_patches, seed=synth.SEED + 7, fixed_patches=args.window_patches)
    n_s = len(w["surf"])
    d_surf, d_pose = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    t_gen = time.perf_counter() - t_gen
    # correspondences (timed: part of the hot path, lidar_odometry.cc:532-538); N > 1: the queries are sharded over the ranks and
    # the gated neighbour lists all-gathered (csrc/match.hip)
    d_pairs, d_pf = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    # (four untimed passes first: the library's scratch buffers are allocated on first use, and the first four calls of a new
    # workload try both orders of the candidate halves twice (csrc/match.hip); then the median of five)
    # (wc_match_pair: both searches side by side on one GPU, one after the other when the matcher is query-sharded)
    for _ in range(4):
        ctx.match_pair_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), d_pairs, n_s, d_pf, n_s, sharded=world > 1)
    ctx.sync()
    t_runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        n_b, n_u = ctx.match_pair_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), d_pairs, n_s, d_pf, n_s, sharded=world > 1)
        t_runs.append(time.perf_counter() - t0)
    t_match = sorted(t_runs)[2]
    # what a query of the fixed-window search touches (the 3-D grid looked at ~860 candidates per query)
    ctx.match_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), False, d_pf, n_s)
    st_fix = ctx.match_stats()
    ctx.match_device(d_surf, d_pose, n_s, d_surf, d_pose, n_s, True, d_pairs, n_s)
    st_sld = ctx.match_stats()
    # N > 1: wc_window_build_sharded - every rank passes the same replicated lists, the library takes this rank's contiguous share of
    # the correspondences and of the IMU triples; every linearisation then ends in ONE all-reduce through the ctx's communicator
    build_args = (d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], False, d_fs, d_fp, d_pf, n_u)
    build_kw = {"sharded": world > 1}
    ctx.window_build(*build_args, **build_kw)  # (first call allocates)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.window_build(*build_args, **build_kw)  # once per solve: interval keys, sort, packed records, pieces, gather lists
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
    lin_call_ms = ctx.timer_stop_ms() / K
    # ... and as the device sees it: 40 linearisations back to back between the two events (wc_window_linearize_timed).  The loop above
    # also times wc_window_linearize's upload of x out of pageable memory, its wait for the mailbox and this interpreter between two
    # calls - none of which the LM loop has between a linearisation's kernels.  The roofline below is priced on the device time.
    lin_ms = ctx.window_linearize_timed(x0, 40)
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
    it_ms = t_solve * 1e3 / iters
    copy = getattr(args, "_copy_gbs", None)
    out = {
        "workload": "C4-like: %d sweeps x %d surfels = %d surfels, %d sample states (%d unknowns)" % (args.window_scans, args.window_patches, n_s, ns, 12 * ns),
        "factors_total": {"binary": n_b, "unary": n_u}, "factors_this_rank": {"binary": nb, "unary": nu, "imu": ni, "pieces": npieces},
        "lm_iterations": summ.iterations, "lm_iters_per_s": round(iters / t_solve, 2), "solve_ms": round(t_solve * 1e3, 3),
        "cost": [summ.initial_cost, summ.final_cost], "termination": summ.termination,
        "linearize_ms": round(lin_ms, 4), "linearize_call_to_call_ms": round(lin_call_ms, 4),
        "assembly_corr_per_s": round(world * (nb + nu) / (lin_ms * 1e-3), 1),
        "assembly_roofline": {"bound": "hbm", "achieved": round(algo / (lin_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(algo / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_linearisation": algo,
                              "frac_call_to_call": round(algo / (lin_call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "timing": "HIP events around 40 linearisations enqueued back to back (wc_window_linearize_timed): k_lin_fused + k_gather and the "
                                        "gap between them; frac_call_to_call = rounds 1 - 4's figure, 20 synchronous wc_window_linearize calls from Python",
                              "frac_of_measured_copy": round(algo / (lin_ms * 1e-3) / 1e9 / copy, 5) if copy else None},
        # one LM iteration = ONE pass over the factor records since round 3 (the candidate's cost comes from a linearisation at the
        # candidate, which an accepted step keeps; round 2 made a cost-only pass and then a linearisation: 2 x the bytes) + the
        # replicated damped solve; the time is the whole iteration (wall clock of the solve / iterations)
        "lm_roofline": {"bound": "hbm", "achieved": round(algo / (it_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_iteration": algo,
                        "passes_over_the_factors": 1, "ms_per_iteration": round(it_ms, 4),
                        "linear_solve": "bias unknowns eliminated by parallel cyclic reduction (12 x 12 super-blocks), dense blocked Cholesky (fp64 MFMA "
                                        "panel steps) on the %d pose unknowns" % (6 * ns)},
        "build_ms": round(t_build * 1e3, 3), "match_s": round(t_match, 4), "match_surfels_per_s": round(2 * n_s / t_match, 1), "generate_s": round(t_gen, 2),
        "match_walk_per_query": {"fixed_window": {k_: round(v_, 1) for k_, v_ in st_fix.items()}, "sliding_window": {k_: round(v_, 1) for k_, v_ in st_sld.items()},
                                 "index": "6-D kd-tree with bounding boxes over [centre, normal / 5 deg] (csrc/match_tree.inc); points_per_query = targets looked at, "
                                          "exact_per_query = fp64 distances (the lane-per-query walk, from 750 k queries on, takes a fp32 first look and sums the "
                                          "survivors; the eight-lanes-per-query walk sums every point of a visited leaf); nodes = node ITEMS (4 / 8 child boxes each).  "
                                          "Round 6: a walk is bounded by the nearest gate-passing candidate as well as by the k-th distance - what "
                                          "KnnSurfelMatcher::Match uses of a neighbour list (cc:24-46); the pair lists are unchanged"},
    }
    # the assembly's OTHER roof (the binding one): fp64 vector issue.  Flops per record by a fixed counting rule - the
    # Gram matrix of the record's row [J r] (upper triangle: 325 / 91 multiply-adds for 24 / 12 unknowns + residual) plus ~450 / ~250
    # for the evaluation (Exp and Jr from one sincos per side, the row right to left, Cauchy corrector); an fma = 2
    flops = nb * (2 * 325 + 450) + nu * (2 * 91 + 250) + ni * 12 * (2 * 37 * 19 + 600)
    out["assembly_flop_roofline"] = {"bound": "fp64 vector", "achieved": round(flops / (lin_ms * 1e-3) / 1e12, 3), "peak": 78.6, "unit": "TFLOP/s",
                                     "frac": round(flops / (lin_ms * 1e-3) / 1e12 / 78.6, 5), "flops_per_binary_record": 2 * 325 + 450,
                                     "flops_per_unary_record": 2 * 91 + 250, "counting_rule": "useful multiply-adds of the formulas, not issued instructions "
                                     "(the kernel issues ~2 300 fp64 instructions per binary record, VALU busy 58 %)"}
    # matcher roofline (48 B feature + 144 B surfel per query and per target, both searches of the window step)
    b_match = 192 * (2 * n_s + n_s + len(w["fix_surf"]))
    out["match_roofline"] = {"bound": "hbm", "achieved": round(b_match / t_match / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b_match / t_match / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_match,
                             "note": "latency-bound gather work (the first gate-passing of the exact 10 nearest in 6-D, through a kd-tree), far from the byte "
                                     "roofline by construction: the window's normals are random, the 10th neighbour lies 2 - 5.7 units away"}
    try:  # the matcher on what a real scanner produces (never quoted without it): surfels on the surfaces of a room, many sweeps deep
        out["match_room_stream"] = bench_match_room(ctx)
    except Exception as e:
        out["match_room_stream"] = {"error": repr(e)}
    if cpu:
        # CPU baseline of the LM step: the single-thread oracle (oracle/window.cc + oracle/match.cc) on the SAME window (BASELINE.md
        # section 3: identical synthetic inputs): both matches, the problem construction and two LM iterations at full size (~20 - 30 s
        # of one core; --cpu-seconds below 5 keeps round 3's 1/20-size sample)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        frac = 1 if args.cpu_seconds >= 5.0 else 20
        ws = w if frac == 1 else synth.surfel_window(args.window_scans, max(1, args.window_patches // frac), seed=synth.SEED + 7,
                                                     fixed_patches=max(1, args.window_patches // frac))
        prm = pyoracle.default_params()
        if frac == 1:
            prm.max_iterations = 2
        with pinned_core() as pc:
            t1 = time.perf_counter()
            pb = pyoracle.match(ws["surf"], ws["pose"], ws["surf"], ws["pose"], True, prm)
            pu = pyoracle.match(ws["surf"], ws["pose"], ws["fix_surf"], ws["fix_pose"], False, prm)
            t_cm = time.perf_counter() - t1
            t1 = time.perf_counter()
            Wc = pyoracle.Window(ws["sample_times"], ws["grav"], False, prm)
            Wc.add_binary(ws["surf"], ws["pose"], pb)
            Wc.add_unary(ws["fix_surf"], ws["fix_pose"], ws["surf"], ws["pose"], pu)
            Wc.add_imu(ws["imu"])
            t_cb = time.perf_counter() - t1
            t1 = time.perf_counter()
            _, sc, _ = Wc.solve(np.zeros(12 * len(ws["sample_times"])))
            t_cs = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(max(1, sc.iterations) / t_cs, 4), "unit": "LM iterations/s", "cores": 1, "kind": "port", "pinned_core": pc.core, "cpu": cpu_info()["model"],
            "matcher_surfels_per_s": round(2 * len(ws["surf"]) / t_cm, 1), "build_s": round(t_cb, 2),
            "sample": ("the SAME window at full size" if frac == 1 else "same window geometry with 1/%d of the surfels" % frac)
                      + " (%d surfels, %d + %d surfel factors, %d unknowns): %d LM iterations in %.1f s; both matches %.1f s"
                      % (len(ws["surf"]), len(pb), len(pu), 12 * len(ws["sample_times"]), sc.iterations, t_cs, t_cm)}
    return out


def ring_allreduce_model_us(nbytes, world):
    """modelled time of a ring all-reduce over xGMI (MI355X_MICROARCH.md: point-to-point links, ~153 GB/s each direction per
    link; a ring moves 2 (N-1)/N of the payload over every rank's one link to its ring neighbour) - a printed estimate, not a
    measurement"""
    if world < 2 or not nbytes:
        return 0.0
    return 2.0 * (world - 1) / world * nbytes / 153e9 * 1e6 + 2 * (world - 1) * 1.5  # + ~1.5 us per hop


def direct_allreduce_model_us(nbytes, world):
    """modelled time of a ONE-SHOT all-reduce on the fully connected xGMI mesh of an 8-GPU MI355X node (7 links per GPU, one to every
    peer): reduce-scatter + all-gather, every rank sending 1/N of the buffer to each peer over its own link at once - (N-1)/N of the
    payload per phase spread over N-1 links -, one hop of latency per phase; a payload of a few cache lines is ONE phase (every rank
    writes its value to every peer: 1 hop).  Link: ~64 GB/s per direction sustained (153 GB/s is the link's two-direction peak);
    hop ~2 us (device-side flag through the fabric).  A printed estimate, not a measurement."""
    if world < 2 or not nbytes:
        return 0.0
    hop, link = 2.0, 64e9
    if nbytes <= 4096:
        return hop + nbytes / link * 1e6
    per_link = nbytes / world  # bytes a rank sends to ONE peer per phase
    return 2.0 * (hop + per_link / link * 1e6)


def multi_gpu_model(result, local_rank):
    """north_star: >= 6 x residual-assembly throughput at 8 GPUs.  No 2+-GPU box is reachable from a 1-GPU run, so this object states
    what a 1-GPU run CAN: the payloads of one linearisation, RCCL's floor for them measured with a world-of-one communicator on the ctx
    stream (enqueue -> completion; no wire), two wire models over xGMI (ring; one-shot on the fully connected mesh) and the MODELLED
    assembly speed-up of rounds 3 - 5's form (one all-reduce behind k_gather, exposed) next to the two-collective form that is the
    default of wc_window_build_sharded since round 6 (16 bytes exposed, the pose corners beside the bias elimination)."""
    from wildcat_slam_amd import lib
    from wildcat_slam_amd import dist as wdist

    out = {"definition": "assembly = k_lin_fused + k_gather of one linearisation (+ the exposed part of its collectives); MODELLED, NOT MEASURED at N > 1 "
                         "(no multi-GPU box is reachable from this run)"}
    probe = {}
    try:
        c = lib.Context(local_rank)
        c.comm_rccl_init(0, 1, lib.rccl_unique_id())
        for ns in (64, 127):
            cnt, cnt2 = wdist.packed_count(ns), wdist.corner_count(ns) - 2
            probe["ns_%d" % ns] = {"one_collective_payload_bytes": 8 * cnt, "two_collective_payload_bytes": [16, 8 * cnt2],
                                   "rccl_world_of_one_us": {"16_bytes": round(c.comm_allreduce_probe(2, 50), 2), "corners": round(c.comm_allreduce_probe(cnt2, 50), 2),
                                                            "one_collective": round(c.comm_allreduce_probe(cnt, 50), 2)}}
        c.comm_rccl_destroy()
        c.close()
    except Exception as e:
        probe["error"] = repr(e)
    out["allreduce"] = probe
    w = result.get("window", {})
    lin_us = 1e3 * w.get("linearize_ms", 0.0)
    if lin_us > 0:
        payload1, payload2 = 8 * wdist.packed_count(127), 8 * (wdist.corner_count(127) - 2)
        imu_floor_us = 14.5  # the IMU family's dependent chains: a launch of their own lasts this long whatever the window
        hide_us = 70.0       # the bias elimination of an iteration at 127 sample states (k_pcr_*: reads IMU blocks only): what the large collective hides behind
        rows = {}
        for n in (2, 4, 8):
            shard = max(lin_us / n, imu_floor_us)
            r = {"shard_assembly_us": round(shard, 1)}
            for name, fn in (("ring", ring_allreduce_model_us), ("direct", direct_allreduce_model_us)):
                one, small, big = fn(payload1, n), fn(16, n), fn(payload2, n)
                exposed_two = small + max(0.0, big - hide_us)
                r[name] = {"one_collective_us": round(one, 1), "sixteen_bytes_us": round(small, 1), "corners_us": round(big, 1),
                           "speedup_one_collective": round(lin_us / (shard + one), 2), "speedup_two_collectives": round(lin_us / (shard + exposed_two), 2)}
            rows[str(n)] = r
        out["c4_window"] = {"assembly_one_gpu_us": round(lin_us, 1), "by_ranks": rows,
                            "which_algorithm": "RCCL picks by size and topology: for 16 bytes its low-latency (LL) protocol - on a fully connected node a direct exchange, "
                                               "one hop -; for 2.3 MB at 8 ranks LL128 / Simple over rings or the direct one-shot path.  The model prints both; "
                                               "a measured 8-GPU run decides (RCCL's small-message floor in practice is nearer 10 - 20 us than the model's 2).",
                            "two_collectives_form": "BUILT (round 6; default of wc_window_build_sharded, development option lm_one_collective for the old form): IMU factors "
                                                    "replicated; {surfel cost} (16 bytes) first - the trust-region decision needs nothing else -; pose corners + pose half "
                                                    "of g (2.3 MB at 127 sample states, was 2.7) on a side stream beside the bias elimination, joined in front of "
                                                    "k_schur_form; tested on 2 - 5 thread-ranks, two processes and two gloo ranks; modelled, not measured"}
    return out


def bench_match_room(ctx):
    """the sliding-window search (KnnSurfelMatcher, knn_surfel_matcher.cc:16-49) on the surfels of a ROOM seen by eight sweeps of a
    spinning scanner (synth.raw_stream, the facade's test stream at 640 k points/s): surfels lie on surfaces, ~280 per cubic metre
    where there are any; the 10th 6-D neighbour is 0.2 units away in the median and 3 - 6 units for the loneliest surfels"""
    from wildcat_slam_amd import records as R, synth

    msgs, _, _ = synth.raw_stream(4.0, pts_per_s=640_000, t_start=1000.0)
    surf = []
    for k in range(0, len(msgs) - 4, 5):
        s, _ = ctx.extract_surfels(synth.concat_points(*msgs[k:k + 5]))
        surf.append(s)
    S = np.concatenate(surf)
    P = np.zeros(len(S), R.POSE)
    P["quat"][:, 0] = 1.0
    n = len(S)
    d_s, d_p, d_pairs = ctx.to_device(S), ctx.to_device(P), ctx.alloc(8 * n)
    ts = []
    for _ in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        m = ctx.match_device(d_s, d_p, n, d_s, d_p, n, True, d_pairs, n)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    b = 192 * 2 * n
    return {"workload": "%d surfels of 8 room sweeps, matched against themselves" % n, "ms_per_search": round(t * 1e3, 4), "pairs": m,
            "queries_per_s": round(n / t, 1), "ms_per_50k_queries": round(t * 1e3 * 50_000 / n, 4),
            "roofline": {"bound": "hbm", "achieved": round(b / t / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / t / 1e9 / HBM_PEAK_GBS, 6)}}


def bench_odometry_step(ctx, args, world, rank, dev, torch, dist, cpu=False):
    """north_star's headline workload, one FULL odometry step (LidarOdometry::AddLidarScan, lidar_odometry.cc:523-566) through the
    C-ABI on a 10-sweep window of 1 M-point sweeps (10 x C2): 9 sweeps are already in the window (extracted, posed; the two oldest
    form the fixed window), the step takes the newest sweep: BuildSurfels -> UpdateSurfelPoses -> 2 x KnnSurfelMatcher -> problem
    construction -> solve -> UpdateSurfelPoses (wildcat_slam_amd/step.py).  N > 1: the SAME step with every stage sharded over the
    ranks (strong scaling: one window, N GPUs) - routed extraction of the newest sweep + gather, query-sharded matcher, sharded
    factors with one all-reduce per linearisation; the time is the slowest rank's."""
    from wildcat_slam_amd import records as R, synth
    from wildcat_slam_amd.step import StepWindow

    K, roots = 10, args.roots
    w = synth.g2_scan_sequence(K, roots, m=32, seed=synth.SEED + 21)
    sw = StepWindow(ctx, w, rank=rank, world=world)
    n_pts, ns = sw.n_pts, sw.ns
    sw.step()
    sw.step()
    reps = 9
    runs, info = [], None
    for _ in range(reps):
        if world > 1:
            dist.barrier()
        T, info, _ = sw.step()
        if world > 1:  # the step lasts as long as its slowest rank
            tt = torch.tensor([T[k_] for k_ in sorted(T)], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            T = dict(zip(sorted(T), [float(v) for v in tt.tolist()]))
        runs.append(T)
    if os.environ.get("WC_BENCH_DEBUG"):
        print("[bench] odometry step, match ms per repetition: " + " ".join("%.2f" % (t["match"] * 1e3) for t in runs), file=sys.stderr)
    runs.sort(key=lambda t: t["total"])
    T = runs[reps // 2]  # the median repetition (a repetition that meets a host hiccup - one in a few dozen - is 2x the others)
    it = max(1, info["iters"])
    # algorithmic bytes of the step: extraction 20 B/pt + 144 B/surfel; pose update ~200 B/surfel R+W per call; matcher
    # 48 B feature + 144 B surfel per query and target per call; assembly (136 / 96 / 128 B per factor) x ONE pass per LM iteration
    # (+ the first linearisation): the candidate's cost comes from a linearisation at the candidate since round 3
    b_ext = 20 * n_pts + 144 * info["new_surfels"]
    b_pose = 2 * 200 * info["sld"]
    b_match = 192 * (2 * info["sld"] + info["sld"] + info["fix"])
    b_asm = (it + 1) * (136 * info["binary"] + 96 * info["unary"] + 128 * info["imu"])
    total_b = b_ext + b_pose + b_match + b_asm
    out = {"workload": "10 x C2 window: %d-point newest sweep; sliding window %d surfels, fixed window %d; %d binary + %d unary + %d IMU factors; %d sample states"
                       % (n_pts, info["sld"], info["fix"], info["binary"], info["unary"], info["imu"], ns),
           "ms_per_step": round(T["total"] * 1e3, 4), "steps_per_s": round(1.0 / T["total"], 2), "points_per_s": round(n_pts / T["total"], 1),
           "stage_ms": {k: round(v * 1e3, 4) for k, v in T.items() if k != "total"}, "lm_iterations": info["iters"], "cost": info["cost"],
           "termination": info["term"],
           "roofline": {"bound": "hbm", "achieved": round(total_b / T["total"] / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(total_b / T["total"] / 1e9 / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes": {"extract": b_ext, "pose_update": b_pose, "match": b_match, "assembly_all_iterations": b_asm}},
           "timing": "wall clock around each C-ABI call, the median of %d repetitions from the same window state; every call is synchronous on return except wc_window_build, whose last device work (records, one copy of lists) ends inside the solve stage" % reps}
    if world == 1:  # the same step with exact_sums = 1 (the arithmetic whose corrections meet 1e-6 against the oracle at this size)
        try:
            ctx.set_exact_sums(True)
            sw.step()
            rx = []
            for _ in range(5):
                Tx, infox, _ = sw.step()
                rx.append(Tx)
            rx.sort(key=lambda t: t["total"])
            Tx = rx[2]
            out["exact_sums"] = {"ms_per_step": round(Tx["total"] * 1e3, 4), "stage_ms": {k: round(v * 1e3, 4) for k, v in Tx.items() if k != "total"},
                                 "lm_iterations": infox["iters"], "default_path_ms_per_step": out["ms_per_step"],
                                 "tolerance_tested": {"exact_sums": 1e-6, "default": 1e-5, "where": "tests/test_step_gpu.py"}}
        except Exception as e:
            out["exact_sums"] = {"error": repr(e)}
        finally:
            ctx.set_exact_sums(False)
    try:  # the fixed-window search alone on this context, for its walk statistics
        n_fix_, n_sld_ = sw.n_fix, info["sld"]
        ctx.match_device(_Ptr(sw.d_surf.ptr + 144 * n_fix_), _Ptr(sw.d_pose.ptr + 56 * n_fix_), n_sld_, sw.d_surf, sw.d_pose, n_fix_, False, sw.d_pu, sw.cap_all)
        out["match_walk_per_query_fixed_window"] = {k_: round(v_, 1) for k_, v_ in ctx.match_stats().items()}
    except Exception as e:
        out["match_walk_per_query_fixed_window"] = {"error": repr(e)}
    if world > 1:
        out["scaling"] = "strong"
        out["n_gpus"] = world
        out["allreduce_bytes_per_linearisation"] = info["allreduce_bytes"]
        out["allreduce_ring_model_us"] = round(ring_allreduce_model_us(info["allreduce_bytes"], world), 1)
        out["collectives"] = ("extraction: one all-to-all of 24-byte point records by root voxel + one all-gather of the new surfels; matcher: one "
                              "all-gather of the gated lists per search; window: one all-reduce per linearisation + one double per candidate cost")
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        # the SAME step at full size (~15 s of one core); --cpu-seconds below 5: 1/10 of the roots per sweep
        full = args.cpu_seconds >= 5.0
        ws = w if full else synth.g2_scan_sequence(K, max(8, roots // 10), m=32, seed=synth.SEED + 21)
        with pinned_core() as pc:  # (the same helper the parity tests of the step compare the GPU path with: tests/test_step_gpu.py)
            ref = pyoracle.odometry_step(ws)
        t_cpu, sl_s, n_fx, pb, pu, sc = ref["seconds"], ref["sld_surf"], ref["n_fix"], ref["pairs_sld"], ref["pairs_fix"], ref["summary"]
        same = None
        if full:  # did the two paths work on the same problem?  (recorded, not asserted: the by-value comparison at this size - surfels,
            # both pair lists, iterations, cost, corrections - is tests/test_step_gpu.py::test_one_rank_step_at_bench_size_against_the_oracle)
            same = (len(sl_s), n_fx, len(pb), len(pu), sc.iterations) == (info["sld"], info["fix"], info["binary"], info["unary"], info["iters"])
        out["cpu_baseline"] = {"value": round(1.0 / t_cpu, 4), "unit": "steps/s", "cores": 1, "kind": "port", "pinned_core": pc.core, "cpu": cpu_info()["model"],
                               "same_counts_and_iterations_as_the_gpu_step": same,
                               "sample": "the same step %s (%d-point sweeps, %d sliding + %d fixed surfels, %d + %d surfel factors, %d LM iterations): %.2f s "
                                         "of single-thread oracle" % ("at FULL size" if full else "at 1/10 scale", len(ws["scans"][-1]), len(sl_s), n_fx, len(pb), len(pu),
                                                                      sc.iterations, t_cpu)}
    return out


def bench_facade_stream(local_rank, cpu):
    """the drop-in class itself: LidarOdometry::AddLidarScan (lidar_odometry.cc:487-605) through libwildcat_odometry.so
    (host/lidar_odometry.cc, INTEGRATION.md route A) on the synthetic room stream of tests/test_facade_gpu.py - 8.2 s of a 32-beam
    scanner at 640 k points/s, messages of 0.1 s, 200 Hz IMU: ms per completed sweep (median behind the first two) with the
    facade's own stage split; beside it the orchestrated CPU oracle (oracle/odometry.cc) on the same messages"""
    from wildcat_slam_amd import lib, synth

    # 8.2 s of stream (16 completed sweeps): the sliding window is 6 s, so from sweep 13 on the FIXED window exists - second search,
    # unary factors, ShrinkToFit (lidar_odometry.cc:228-250) are inside the timed sweeps (tests/test_facade_gpu.py
    # compares the same stream with the orchestrated oracle)
    msgs, imu, _ = synth.raw_stream(8.2, pts_per_s=640_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)

    def drive(odo, stage=None):
        k, times, before = 0, [], 0
        for m in msgs:
            if len(m) == 0:
                continue
            t_end = m["time"][-1]
            while k < len(imu["t"]) and imu["t"][k] <= t_end + 0.02:
                odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
                k += 1
            t0 = time.perf_counter()
            odo.add_scan(m)
            dt = time.perf_counter() - t0
            if odo.sweeps() > before:
                before = odo.sweeps()
                times.append((dt, stage() if stage else None, dict(odo.stats())))
        return times

    odo = lib.Odometry(local_rank)
    times = drive(odo, odo.stage_ms)
    fast, exact = odo.extract_paths()
    odo.close()
    tail = times[2:]
    ts = np.array([t for t, _, _ in tail])
    med = {k_: round(float(np.median([st[k_] for _, st, _ in tail])), 4) for k_ in tail[0][1]}
    out = {"workload": "room stream, %d completed sweeps of ~%d points; sliding window %d .. %d surfels, fixed window up to %d" % (
               len(times), int(np.mean([len(m) for m in msgs])) * 5, int(min(s_["sld_surfels"] for _, _, s_ in tail)), int(max(s_["sld_surfels"] for _, _, s_ in tail)),
               int(max(s_["fix_surfels"] for _, _, s_ in tail))),
           "ms_per_sweep_median": round(float(np.median(ts)) * 1e3, 4), "ms_per_sweep_max": round(float(ts.max()) * 1e3, 4),
           "ms_per_sweep": [round(float(t) * 1e3, 3) for t in ts], "lm_iterations_per_sweep": [int(st["lm_iterations"]) for _, st, _ in tail],
           "ms_per_lm_iteration_median": round(float(np.median([st["solve"] / max(st["lm_iterations"], 1.0) for _, st, _ in tail])), 4),
           "fix_surfels_per_sweep": [int(s_["fix_surfels"]) for _, _, s_ in tail], "unary_per_sweep": [int(s_["unary"]) for _, _, s_ in tail],
           "stage_ms_median": med, "sweeps_on_the_default_extraction_path": fast, "sweeps_on_the_exact_path": exact,
           "timing": "wall clock around LidarOdometry::AddLidarScan for the message that completes a sweep; stages by the facade's own clock"}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # test infrastructure, used here ONLY as the timed CPU baseline

        with pinned_core() as pc:
            oo = pyoracle.Odometry()
            ot = drive(oo)
            oo.close()
        otail = np.array([t for t, _, _ in ot[2:]])
        out["cpu_baseline"] = {"value": round(float(np.median(otail)) * 1e3, 3), "unit": "ms per sweep (median)", "cores": 1, "kind": "port", "pinned_core": pc.core,
                               "sample": "the same %d sweeps through the orchestrated oracle: %.1f s" % (len(ot), float(sum(t for t, _, _ in ot)))}
    return out


def ctx_ptr(buf):
    import ctypes

    return ctypes.c_void_p(buf.ptr)


def ctx_size(n):
    import ctypes

    return ctypes.c_size_t(int(n))


if __name__ == "__main__":
    main()
