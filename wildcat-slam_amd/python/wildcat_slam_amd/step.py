"""One full odometry step through the C-ABI - the hot block of LidarOdometry::AddLidarScan (lidar_odometry.cc:523-566):
BuildSurfels -> UpdateSurfelPoses -> 2 x KnnSurfelMatcher -> problem construction -> solve -> UpdateSurfelPoses - on a window that
is resident in HBM, for ONE rank or for the N ranks of a multi-GPU job.  bench.py times it, the tests compare the
N-rank step with the 1-rank step; the window state is replicated, so every rank ends with the same sample-state corrections.

N ranks (sharded=True; every call below is a collective of the ctx's communicator):
  extraction   the newest sweep arrives as time-contiguous slices, one per rank: wc_extract_surfels_sharded (one all-to-all of
               24-byte point records by root voxel) + wc_gather_surfels (all-gather + merge: the replicated window gets the list
               the unsharded call returns)
  matcher      wc_match_pair_sharded: queries sharded, one all-gather of the gated lists per search
  window       wc_window_build_sharded + wc_window_solve: correspondences and IMU triples sharded, ONE all-reduce per
               linearisation (0.76 MB at 64 sample states)
"""
import ctypes as C
import time

import numpy as np

from . import dist as wdist
from . import records as R


class _Ptr:
    def __init__(self, ptr):
        self.ptr = int(ptr)

    def __bool__(self):
        return self.ptr != 0


class StepWindow:
    """device-resident window in front of the step: the surfels of sweeps 0 .. K-2 (extracted, posed; the first `n_fix` form the
    fixed window), the newest sweep's points, the IMU states and sample times"""

    def __init__(self, ctx, w, rank=0, world=1, keep_ids=False):
        """keep_ids: the surfel ids of every sweep are kept next to the surfels (d_ids; the parity tests identify surfels by them)"""
        self.ctx, self.w, self.rank, self.world = ctx, w, rank, world
        scans, imu = w["scans"], w["imu"]
        self.imu = imu
        self.d_imu = ctx.to_device(imu)
        self.newest = scans[-1]
        self.n_pts = len(self.newest)
        roots_cap = sum(len(s) for s in scans) * 3 // 20 + 4096
        self.cap_all = roots_cap
        self.d_surf, self.d_pose, self.d_inb = ctx.alloc(144 * roots_cap), ctx.alloc(56 * roots_cap), ctx.alloc(roots_cap)
        ctx._ck(ctx.lib.wc_memset(ctx.h, C.c_void_p(self.d_inb.ptr), 0, C.c_size_t(roots_cap)))
        self.d_ids = ctx.alloc(16 * roots_cap) if keep_ids else None
        counts, n_have = [], 0
        for s in scans[:-1]:  # the window before the step (every rank extracts it: replicated state, outside the step)
            d = ctx.to_device(s)
            ctx.extract_enqueue(ctx.points_desc(d, len(s)), _Ptr(self.d_surf.ptr + 144 * n_have), _Ptr(self.d_ids.ptr + 16 * n_have) if keep_ids else None,
                                roots_cap - n_have, float(s["time"][0]), float(s["time"][-1]))
            m = ctx.extract_finish()
            counts.append(m)
            n_have += m
        ctx.update_surfel_poses(self.d_imu, len(imu), self.d_surf, self.d_pose, self.d_inb, n_have)
        self.n_have, self.n_fix = n_have, counts[0] + counts[1]  # the two oldest sweeps: fixed window
        self.counts = counts
        self.d_pb, self.d_pu = ctx.alloc(8 * roots_cap), ctx.alloc(8 * roots_cap)
        # the newest sweep: all of it on one rank, this rank's time-contiguous slice on several
        lo, cnt = wdist.shard_range(self.n_pts, rank, world)
        self.slice_n = cnt
        self.d_new = ctx.to_device(self.newest[lo:lo + cnt] if world > 1 else self.newest)
        self.t_lo, self.t_hi = float(self.newest["time"][0]), float(self.newest["time"][-1])
        cap_new = (3 * self.n_pts) // 20 + 1
        self.cap_new = cap_new
        self.d_loc, self.d_loc_ids = ctx.alloc(144 * cap_new), ctx.alloc(16 * cap_new)  # this rank's surfels of the newest sweep
        # a copy of the window so that every repetition starts from the same state
        self.keep = (ctx.alloc(144 * n_have), ctx.alloc(56 * n_have))
        ctx._ck(ctx.lib.wc_d2d(ctx.h, C.c_void_p(self.keep[0].ptr), C.c_void_p(self.d_surf.ptr), C.c_size_t(144 * n_have)))
        ctx._ck(ctx.lib.wc_d2d(ctx.h, C.c_void_p(self.keep[1].ptr), C.c_void_p(self.d_pose.ptr), C.c_size_t(56 * n_have)))
        self.ns = len(w["sample_times"])

    def reset(self):
        ctx, n = self.ctx, self.n_have
        ctx._ck(ctx.lib.wc_d2d(ctx.h, C.c_void_p(self.d_surf.ptr), C.c_void_p(self.keep[0].ptr), C.c_size_t(144 * n)))
        ctx._ck(ctx.lib.wc_d2d(ctx.h, C.c_void_p(self.d_pose.ptr), C.c_void_p(self.keep[1].ptr), C.c_size_t(56 * n)))
        ctx._ck(ctx.lib.wc_memset(ctx.h, C.c_void_p(self.d_inb.ptr + n), 0, C.c_size_t(self.cap_all - n)))
        ctx.sync()

    def step(self):
        """-> (stage wall times [s], info dict, sample-state corrections x)"""
        ctx, w = self.ctx, self.w
        sharded = self.world > 1
        self.reset()
        T = {}
        t0 = time.perf_counter()
        dst = _Ptr(self.d_surf.ptr + 144 * self.n_have)
        dst_ids = _Ptr(self.d_ids.ptr + 16 * self.n_have) if self.d_ids else None
        if sharded:
            _, _, m_loc, _ = ctx.extract_surfels_sharded(self.d_new, self.slice_n, self.t_lo, self.t_hi, out=(self.d_loc, self.d_loc_ids, self.cap_new))
            m = ctx.gather_surfels_device(self.d_loc, self.d_loc_ids, m_loc, dst, dst_ids, self.cap_all - self.n_have)
        else:
            ctx.extract_enqueue(ctx.points_desc(self.d_new, self.n_pts), dst, dst_ids, self.cap_all - self.n_have, self.t_lo, self.t_hi)
            m = ctx.extract_finish()
        T["extract"] = time.perf_counter() - t0
        n_all = self.n_have + m
        n_fix, n_sld = self.n_fix, n_all - self.n_fix
        sld_s, sld_p, sld_b = _Ptr(self.d_surf.ptr + 144 * n_fix), _Ptr(self.d_pose.ptr + 56 * n_fix), _Ptr(self.d_inb.ptr + n_fix)
        t1 = time.perf_counter()
        ctx.update_surfel_poses(self.d_imu, len(self.imu), sld_s, sld_p, sld_b, n_sld)
        T["pose_update"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        nb, nu = ctx.match_pair_device(sld_s, sld_p, n_sld, self.d_surf, self.d_pose, n_fix, self.d_pb, self.cap_all, self.d_pu, self.cap_all, sharded=sharded)
        T["match"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        ctx.window_build(sld_s, sld_p, self.d_pb, nb, self.imu, w["sample_times"], w["grav"], False, self.d_surf, self.d_pose, self.d_pu, nu, sharded=sharded)
        T["build"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        x, summ, _ = ctx.window_solve(np.zeros(12 * self.ns))
        T["solve"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        ctx.update_surfel_poses(self.d_imu, len(self.imu), sld_s, sld_p, sld_b, n_sld)  # (the IMU poses would carry the B-spline correction)
        ctx.sync()
        T["pose_update2"] = time.perf_counter() - t1
        T["total"] = time.perf_counter() - t0
        info = dict(new_surfels=m, sld=n_sld, fix=n_fix, binary=nb, unary=nu, iters=summ.iterations, cost=[summ.initial_cost, summ.final_cost],
                    term=summ.termination, imu=max(0, len(self.imu) - 2), allreduce_bytes=ctx.window_reduce_bytes(), counts=self.counts + [m])
        return T, info, x
