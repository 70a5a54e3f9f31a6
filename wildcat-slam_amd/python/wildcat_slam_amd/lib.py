"""ctypes binding of libwildcat_hip.so (include/wildcat_hip.h).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible, every compute entry
point raises.  (`load()` alone works without a GPU so that the CPU test-suite can check the exported symbols.)
"""
import ctypes as C
import os
import re
import weakref

import numpy as np

from . import records as R

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "csrc")
_SO = os.path.join(_CSRC, "libwildcat_hip.so")
_INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include")
_LIB = None

WC_OK, WC_ERR_CAPACITY = 0, 1


class WildcatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wildcat_hip error {code}: {msg}")
        self.code = code


def so_path():
    return _SO


def load():
    """Load libwildcat_hip.so; raises if it has not been built (see __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError(f"{_SO} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(_SO)
        _LIB.wc_version.restype = C.c_char_p
        _LIB.wc_last_error.restype = C.c_char_p
    return _LIB


def declared_symbols():
    """Every function name declared in include/wildcat_hip.h."""
    txt = open(os.path.join(_INCLUDE, "wildcat_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wc_[a-z0-9_]+)\s*\(", txt)))


def default_params():
    p = R.Params()
    load().wc_params_default(C.byref(p))
    return p


def rccl_unique_id():
    """128 opaque bytes from ncclGetUniqueId (rank 0 creates them, the launcher distributes them)"""
    buf = C.create_string_buffer(128)
    rc = load().wc_comm_rccl_unique_id(buf)
    if rc != WC_OK:
        raise WildcatError(rc, "wc_comm_rccl_unique_id failed (librccl.so not loadable?)")
    return bytes(buf.raw)


def route_owner(keys_xyz, world):
    """owner rank of every root voxel index (n x 3 int32) - wc_route_owner, needs no GPU"""
    keys_xyz = np.asarray(keys_xyz, np.int64).reshape(-1, 3)
    uniq, inv = np.unique(keys_xyz, axis=0, return_inverse=True)
    f = load().wc_route_owner
    own = np.array([f(int(k[0]), int(k[1]), int(k[2]), int(world)) for k in uniq], np.int32)
    return own[inv.reshape(-1)]


class DeviceBuffer:
    """A block of HBM owned through the C-ABI (wc_dev_alloc / wc_dev_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p(0)
        ctx._ck(ctx.lib.wc_dev_alloc(ctx.h, C.c_size_t(max(self.nbytes, 1)), C.byref(p)))
        self.ptr = p.value
        ctx._bufs.add(self)  # Context.close() frees what is still alive (wc_dev_free needs the live ctx)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.lib.wc_h2d(self.ctx.h, C.c_void_p(self.ptr), R.ptr(arr), C.c_size_t(arr.nbytes)))
        return self

    def download(self, dtype, count):
        out = np.zeros(count, dtype)
        self.ctx._ck(self.ctx.lib.wc_d2h(self.ctx.h, R.ptr(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.wc_dev_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """wc_ctx wrapper: one GPU, one stream."""

    def __init__(self, device=0, params=None):
        self.lib = load()
        self.params = params or default_params()
        self._bufs = weakref.WeakSet()
        h = C.c_void_p(0)
        rc = self.lib.wc_ctx_create(C.byref(self.params), C.c_int(device), C.byref(h))
        if rc != WC_OK:
            raise WildcatError(rc, "wc_ctx_create failed (no gfx950 device visible?)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            for b in list(self._bufs):  # buffers outliving the context would leak (wc_dev_free(NULL, p) is an error)
                b.free()
            self.lib.wc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != WC_OK:
            raise WildcatError(rc, self.lib.wc_last_error(self.h).decode())

    def set_params(self, params):
        self.params = params
        self._ck(self.lib.wc_ctx_set_params(self.h, C.byref(params)))

    def selftest_factor32(self, a, variant, reps=5):
        """-> (L, L^-1, shader clocks, ok) of the 32 x 32 SPD matrix a by the solve's diagonal-block kernel (wc_selftest_factor32)"""
        a = np.ascontiguousarray(a, np.float64)
        L, X = np.zeros((32, 32)), np.zeros((32, 32))
        clk = (C.c_longlong * 2)()
        self._ck(self.lib.wc_selftest_factor32(self.h, C.c_int(variant), C.c_int(reps), R.ptr(a), R.ptr(L), R.ptr(X), clk))
        return L, X, int(clk[0]), bool(clk[1])

    def set_dev_option(self, name, value):
        """a development option of this context (include/wildcat_hip.h: wc_ctx_set_dev_option)"""
        self._ck(self.lib.wc_ctx_set_dev_option(self.h, name.encode(), C.c_int(int(value))))

    def set_stream(self, stream_ptr):
        self._ck(self.lib.wc_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    def sync(self):
        self._ck(self.lib.wc_sync(self.h))

    def download_raw(self, d_ptr, nbytes):
        """bytes at a raw device pointer (e.g. the buffer handed to an all-reduce callback) -> uint8 array"""
        out = np.zeros(int(nbytes), np.uint8)
        self._ck(self.lib.wc_d2h(self.h, R.ptr(out), C.c_void_p(d_ptr), C.c_size_t(out.nbytes)))
        return out

    def upload_raw(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.lib.wc_h2d(self.h, C.c_void_p(d_ptr), R.ptr(arr), C.c_size_t(arr.nbytes)))

    def timer_start(self):
        self._ck(self.lib.wc_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float(0)
        self._ck(self.lib.wc_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- extraction ---------------------------------------------------------------------------------------------
    def points_desc(self, d_points, n):
        """descriptor for a device-resident array of 48-byte hilti_ros::Point records"""
        return R.Points(d_points.ptr, d_points.ptr + 24, 48, 48, n)

    def voxel_keys(self, points):
        d = self.to_device(points)
        out = self.alloc(12 * len(points))
        desc = self.points_desc(d, len(points))
        self._ck(self.lib.wc_voxel_keys(self.h, C.byref(desc), C.c_void_p(out.ptr)))
        return out.download(np.int32, 3 * len(points)).reshape(-1, 3)

    def extract_enqueue(self, desc, d_out, d_ids, cap, t_lo=1.0, t_hi=0.0):
        self._ck(self.lib.wc_extract_surfels_enqueue(self.h, C.byref(desc), C.c_double(t_lo), C.c_double(t_hi), C.c_void_p(d_out.ptr),
                                                     C.c_void_p(d_ids.ptr if d_ids else 0), C.c_uint64(cap)))

    def extract_finish(self):
        n = C.c_uint64(0)
        self._ck(self.lib.wc_extract_surfels_finish(self.h, C.byref(n)))
        return int(n.value)

    def extract_batch_prepare(self, jobs):
        """jobs: list of (desc, d_out, d_ids or None, cap, t_lo, t_hi) -> (enqueue(), finish() -> [n_surfels per sweep]): K sweeps
        through one launch chain (wc_extract_surfels_batch_*)"""
        K = len(jobs)
        arr = (R.SweepJob * K)()
        for k, (desc, d_out, d_ids, cap, t_lo, t_hi) in enumerate(jobs):
            arr[k].pts, arr[k].t_lo, arr[k].t_hi = desc, t_lo, t_hi
            arr[k].d_out, arr[k].d_ids, arr[k].cap = d_out.ptr, (d_ids.ptr if d_ids else 0), cap
        counts = (C.c_uint64 * K)()
        enq, fin, ck, h = self.lib.wc_extract_surfels_batch_enqueue, self.lib.wc_extract_surfels_batch_finish, self._ck, self.h

        def enqueue():
            ck(enq(h, arr, C.c_int(K)))

        def finish():
            ck(fin(h, counts, C.c_int(K)))
            return [int(c) for c in counts]

        return enqueue, finish

    def prepare_extract(self, desc, d_out, d_ids, cap, t_lo=1.0, t_hi=0.0):
        """the same enqueue / finish pair with the ctypes argument objects built once: a sweep takes ~80 us, building them
        anew for every call is several us of host turn-around.  -> (enqueue(), finish() -> n_surfels)"""
        args = (self.h, C.byref(desc), C.c_double(t_lo), C.c_double(t_hi), C.c_void_p(d_out.ptr), C.c_void_p(d_ids.ptr if d_ids else 0),
                C.c_uint64(cap))
        n = C.c_uint64(0)
        n_ref = C.byref(n)
        f_enq, f_fin, h, ck = self.lib.wc_extract_surfels_enqueue, self.lib.wc_extract_surfels_finish, self.h, self._ck

        def enqueue():
            rc = f_enq(*args)
            if rc:
                ck(rc)

        def finish():
            rc = f_fin(h, n_ref)
            if rc:
                ck(rc)
            return n.value

        return enqueue, finish

    def set_exact_sums(self, exact):
        """extraction arithmetic: False (default) = order-independent integer moments, True = the reference's summation order"""
        self.params.exact_sums = 1 if exact else 0
        self.set_params(self.params)

    def extract_path_info(self):
        """-> dict(fast=the last sweep was completed by the fast path, fallbacks=sweeps handed to the exact path so far, flags,
        long_lists=the last fast sweep had long record lists: the next one merges them first (k_fx_merge))"""
        w = (C.c_uint32 * 64)()
        self._ck(self.lib.wc_debug_status(self.h, w))
        return dict(fast=bool(w[61]), fallbacks=int(w[60]), flags=int(w[62]), why=int(w[63]), long_lists=bool(w[59]))

    def extract_profile(self, enable=True):
        """True / 1: HIP events after every kernel group of the stage (each event costs ~5 us of stream time); 2: only the first
        and the last one (extract_stage_ms then reports the whole stage under "point_sort"); False: off"""
        self._ck(self.lib.wc_extract_profile(self.h, C.c_int(int(enable))))

    def extract_stage_ms(self):
        ms = (C.c_float * 5)()
        self._ck(self.lib.wc_extract_stage_ms(self.h, ms))
        return dict(zip(("init", "point_sort", "roots_stream", "roots_emit", "slot_order"), [float(v) for v in ms]))

    def extract_surfels(self, points, hint=True, cap=None):
        """host convenience: upload POINT array, extract, download -> (surfels, ids).  hint: True = the sweep's own time range,
        False = none (the library reads it back), (t_lo, t_hi) = explicit range"""
        n = len(points)
        assert points.dtype.itemsize == 48, "POINT records are 48 bytes (np.concatenate packs the dtype: use synth.concat_points)"
        cap = cap or max(1024, (3 * n) // 20 + 1)
        d_pts = self.to_device(points) if n else self.alloc(48)
        d_out, d_ids = self.alloc(cap * 144), self.alloc(cap * 16)
        desc = self.points_desc(d_pts, n)
        if isinstance(hint, tuple):
            t_lo, t_hi = hint
        elif hint and n:
            t_lo, t_hi = float(points["time"][0]), float(points["time"][-1])
        else:
            t_lo, t_hi = 1.0, 0.0
        self.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi)
        m = self.extract_finish()
        return d_out.download(R.SURFEL, m), d_ids.download(R.SURFEL_ID, m)

    # ---- one cloud over several GPUs (csrc/route.hip) --------------------------------------------------------------------
    def set_comm(self, comm):
        """comm: object with .rank, .world and methods allreduce(ptr, count), alltoallv(send_ptr, send_bytes, recv_ptr, recv_bytes),
        allgatherv(send_ptr, send_bytes, recv_ptr, recv_bytes) on raw device pointers (see dist.py); None removes it"""
        if comm is None:
            self._comm = None
            self._ck(self.lib.wc_ctx_set_comm(self.h, C.c_void_p(0)))
            return
        world = comm.world

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # pragma: no cover
                    print("communicator callback failed:", repr(e))
                    return 1

            return wrapped

        ar = R.COMM_ALLREDUCE(guard(lambda user, p, n: comm.allreduce(p, n)))
        a2a = R.COMM_ALLTOALLV(guard(lambda user, s, sb, r, rb: comm.alltoallv(s, [sb[i] for i in range(world)], r, [rb[i] for i in range(world)])))
        ag = R.COMM_ALLGATHERV(guard(lambda user, s, n, r, rb: comm.allgatherv(s, n, r, [rb[i] for i in range(world)])))
        c = R.Comm(None, comm.rank, world, ar, a2a, ag, 0)
        self._comm = (c, ar, a2a, ag, comm)  # keep the trampolines alive
        self._ck(self.lib.wc_ctx_set_comm(self.h, C.byref(c)))

    def comm_rccl_init(self, rank, world, unique_id):
        """the in-library RCCL communicator (csrc/comm.hip); unique_id: the 128 bytes of rccl_unique_id() of rank 0"""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self.lib.wc_comm_rccl_init(self.h, C.c_int(rank), C.c_int(world), buf))

    def comm_allreduce_probe(self, count, reps=50):
        """microseconds per all-reduce of `count` doubles through the installed communicator (wc_comm_allreduce_probe)"""
        us = C.c_double(0)
        self._ck(self.lib.wc_comm_allreduce_probe(self.h, C.c_uint64(int(count)), C.c_int(int(reps)), C.byref(us)))
        return float(us.value)

    def comm_rccl_destroy(self):
        self._ck(self.lib.wc_comm_rccl_destroy(self.h))

    def route_partition(self, d_points, n, world):
        """-> (DeviceBuffer of wc_route_point[n], counts[world])"""
        d_send = self.alloc(24 * max(n, 1))
        counts = (C.c_uint64 * world)()
        desc = self.points_desc(d_points, n)
        self._ck(self.lib.wc_route_partition(self.h, C.byref(desc), C.c_int(world), C.c_void_p(d_send.ptr), counts))
        return d_send, np.array(list(counts), np.uint64)

    def route_desc(self, d_route_points, n):
        return R.Points(d_route_points.ptr, d_route_points.ptr + 16, 24, 24, n)

    def extract_surfels_sharded(self, d_points, n, t_lo, t_hi, cap=None, out=None):
        """this rank's slice (device POINT array) -> (surfels, ids, count, points owned) of the voxels this rank owns;
        out = (d_out, d_ids, cap) reuses buffers"""
        if out is not None:
            d_out, d_ids, cap = out
        else:
            cap = cap or max(1024, (3 * n) // 20 + 1) * 4
            d_out, d_ids = self.alloc(cap * 144), self.alloc(cap * 16)
        desc = self.points_desc(d_points, n)
        m, owned = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.lib.wc_extract_surfels_sharded(self.h, C.byref(desc), C.c_double(t_lo), C.c_double(t_hi), C.c_void_p(d_out.ptr),
                                                     C.c_void_p(d_ids.ptr), C.c_uint64(cap), C.byref(m), C.byref(owned)))
        return d_out, d_ids, int(m.value), int(owned.value)

    def merge_surfels(self, lists, id_lists):
        """host convenience: k sorted SURFEL arrays (+ ids) -> merged (surfels, ids)"""
        counts = (C.c_uint64 * len(lists))(*[len(a) for a in lists])
        n = sum(len(a) for a in lists)
        d_in, d_ids = self.to_device(np.concatenate(lists)), self.to_device(np.concatenate(id_lists))
        d_out, d_oid = self.alloc(144 * max(n, 1)), self.alloc(16 * max(n, 1))
        self._ck(self.lib.wc_merge_surfels(self.h, C.c_void_p(d_in.ptr), C.c_void_p(d_ids.ptr), counts, C.c_int(len(lists)), C.c_void_p(d_out.ptr),
                                           C.c_void_p(d_oid.ptr)))
        return d_out.download(R.SURFEL, n), d_oid.download(R.SURFEL_ID, n)

    def gather_surfels_device(self, d_local, d_local_ids, n_local, d_out, d_out_ids, cap):
        """wc_gather_surfels into caller buffers (device) -> number of surfels of the merged, replicated list"""
        m = C.c_uint64(0)
        self._ck(self.lib.wc_gather_surfels(self.h, C.c_void_p(d_local.ptr), C.c_void_p(d_local_ids.ptr), C.c_uint64(n_local), C.c_void_p(d_out.ptr),
                                            C.c_void_p(d_out_ids.ptr if d_out_ids else 0), C.c_uint64(cap), C.byref(m)))
        return int(m.value)

    def gather_surfels(self, d_local, d_local_ids, n_local, cap):
        d_out, d_oid = self.alloc(144 * max(cap, 1)), self.alloc(16 * max(cap, 1))
        m = C.c_uint64(0)
        self._ck(self.lib.wc_gather_surfels(self.h, C.c_void_p(d_local.ptr), C.c_void_p(d_local_ids.ptr), C.c_uint64(n_local), C.c_void_p(d_out.ptr),
                                            C.c_void_p(d_oid.ptr), C.c_uint64(cap), C.byref(m)))
        return d_out.download(R.SURFEL, int(m.value)), d_oid.download(R.SURFEL_ID, int(m.value))

    # ---- sweep preparation (row f-1) -------------------------------------------------------------------------------------
    def prefilter_points(self, points, ext_quat, ext_t, min_range, max_range, blind_min, blind_max):
        n = len(points)
        d_in, d_out = self.to_device(points), self.alloc(48 * max(n, 1))
        m = C.c_uint64(0)
        v = lambda a: R.ptr(np.ascontiguousarray(a, np.float64))  # noqa: E731
        self._ck(self.lib.wc_prefilter_points(self.h, C.c_void_p(d_in.ptr), C.c_uint64(n), v(ext_quat), v(ext_t), C.c_double(min_range),
                                              C.c_double(max_range), v(blind_min), v(blind_max), C.c_void_p(d_out.ptr), C.c_uint64(n), C.byref(m)))
        return d_out.download(R.POINT, int(m.value))

    def prefilter_points_checked(self, points, ext_quat, ext_t, min_range, max_range, blind_min, blind_max, prev_time=-np.inf):
        """-> (survivors, their stamps as the library packed them, the reference's CHECK at lidar_odometry.cc:491 held)"""
        n = len(points)
        d_in, d_out, d_t = self.to_device(points), self.alloc(48 * max(n, 1)), self.alloc(8 * max(n, 1))
        m, mono = C.c_uint64(0), C.c_int(1)
        v = lambda a: R.ptr(np.ascontiguousarray(a, np.float64))  # noqa: E731
        self._ck(self.lib.wc_prefilter_points_checked(self.h, C.c_void_p(d_in.ptr), C.c_uint64(n), v(ext_quat), v(ext_t), C.c_double(min_range),
                                                      C.c_double(max_range), v(blind_min), v(blind_max), C.c_void_p(d_out.ptr), C.c_uint64(n),
                                                      C.byref(m), C.c_double(prev_time), C.c_void_p(d_t.ptr), C.byref(mono)))
        k = int(m.value)
        return d_out.download(R.POINT, k), d_t.download(np.float64, k), bool(mono.value)

    def undistort_sweep_packed(self, points, imu, keep_on_device=False):
        """-> (xyz float32 (n, 3), time float64 (n,)): the 20 bytes per point the extraction reads"""
        n = len(points)
        d_in, d_imu = self.to_device(points), self.to_device(imu)
        d_xyz, d_t = self.alloc(12 * max(n, 1)), self.alloc(8 * max(n, 1))
        self._ck(self.lib.wc_undistort_sweep_packed(self.h, C.c_void_p(d_in.ptr), C.c_uint64(n), C.c_void_p(d_imu.ptr), C.c_uint64(len(imu)),
                                                    C.c_void_p(d_xyz.ptr), C.c_void_p(d_t.ptr)))
        if keep_on_device:
            return d_xyz, d_t
        return d_xyz.download(np.float32, 3 * n).reshape(-1, 3), d_t.download(np.float64, n)

    def undistort_sweep(self, points, imu):
        n = len(points)
        d_in, d_out, d_imu = self.to_device(points), self.alloc(48 * max(n, 1)), self.to_device(imu)
        self._ck(self.lib.wc_undistort_sweep(self.h, C.c_void_p(d_in.ptr), C.c_uint64(n), C.c_void_p(d_imu.ptr), C.c_uint64(len(imu)), C.c_void_p(d_out.ptr)))
        return d_out.download(R.POINT, n)

    # ---- pose update / window problem ------------------------------------------------------------------------------
    def update_surfel_poses(self, d_imu, n_imu, d_surf, d_pose, d_in_body, n):
        self._ck(self.lib.wc_update_surfel_poses(self.h, C.c_void_p(d_imu.ptr), C.c_uint64(n_imu), C.c_void_p(d_surf.ptr), C.c_void_p(d_pose.ptr),
                                                 C.c_void_p(d_in_body.ptr), C.c_uint64(n)))

    def window_build(self, d_surf, d_pose, d_pairs, n_pairs, imu, sample_times, grav, fix_first_pos, d_fix_surf=None, d_fix_pose=None,
                     d_pairs_fix=None, n_pairs_fix=0, sharded=False):
        """imu: host IMU_STATE array (or None); sample_times/grav: host arrays.  sharded: wc_window_build_sharded - a collective of
        the ctx's communicator, every rank passes the SAME replicated arguments and the library takes this rank's share"""
        st = np.ascontiguousarray(sample_times, np.float64)
        gr = np.ascontiguousarray(grav, np.float64)
        self._ns = len(st)
        null = C.c_void_p(0)
        fn = self.lib.wc_window_build_sharded if sharded else self.lib.wc_window_build
        self._ck(fn(
            self.h, C.c_void_p(d_surf.ptr), C.c_void_p(d_pose.ptr), C.c_void_p(d_pairs.ptr) if d_pairs else null, C.c_uint64(n_pairs),
            C.c_void_p(d_fix_surf.ptr) if d_fix_surf else null, C.c_void_p(d_fix_pose.ptr) if d_fix_pose else null,
            C.c_void_p(d_pairs_fix.ptr) if d_pairs_fix else null, C.c_uint64(n_pairs_fix),
            R.ptr(imu) if imu is not None and len(imu) else null, C.c_uint64(len(imu) if imu is not None else 0), R.ptr(st), C.c_uint64(len(st)),
            R.ptr(gr), C.c_int(int(fix_first_pos))))

    def window_reduce_bytes(self):
        """bytes one linearisation's all-reduce carries (0: the problem is not sharded)"""
        self.lib.wc_window_reduce_bytes.restype = C.c_uint64
        return int(self.lib.wc_window_reduce_bytes(self.h))

    def window_counts(self):
        c = (C.c_uint64 * 4)()
        self._ck(self.lib.wc_window_counts(self.h, c))
        return list(c)

    def window_evaluate(self, x, want_residuals=False):
        x = np.ascontiguousarray(x, np.float64)
        cost = C.c_double(0)
        if want_residuals:
            nb, nu, ni, _ = self.window_counts()
            d_res = self.alloc(8 * (nb + nu + 12 * ni))
            self._ck(self.lib.wc_window_evaluate(self.h, R.ptr(x), C.byref(cost), C.c_void_p(d_res.ptr)))
            return cost.value, d_res.download(np.float64, nb + nu + 12 * ni)
        self._ck(self.lib.wc_window_evaluate(self.h, R.ptr(x), C.byref(cost), C.c_void_p(0)))
        return cost.value

    def window_linearize(self, x):
        x = np.ascontiguousarray(x, np.float64)
        n = 12 * self._ns
        d_H, d_g = self.alloc(8 * n * n), self.alloc(8 * n)
        cost = C.c_double(0)
        self._ck(self.lib.wc_window_linearize(self.h, R.ptr(x), C.c_void_p(d_H.ptr), C.c_void_p(d_g.ptr), C.byref(cost)))
        return d_H.download(np.float64, n * n).reshape(n, n), d_g.download(np.float64, n), cost.value

    def window_linearize_only(self, x):
        """linearise without copying H / g out (benchmarks)"""
        x = np.ascontiguousarray(x, np.float64)
        cost = C.c_double(0)
        self._ck(self.lib.wc_window_linearize(self.h, R.ptr(x), C.c_void_p(0), C.c_void_p(0), C.byref(cost)))
        return cost.value

    def window_linearize_timed(self, x, reps=20):
        """device ms of one linearisation: `reps` of them back to back between two HIP events"""
        x = np.ascontiguousarray(x, np.float64)
        ms = C.c_float(0)
        self._ck(self.lib.wc_window_linearize_timed(self.h, R.ptr(x), int(reps), C.byref(ms)))
        return ms.value

    def window_solve(self, x):
        x = np.ascontiguousarray(x, np.float64).copy()
        s = R.SolveSummary()
        first = np.zeros(len(x))
        self._ck(self.lib.wc_window_solve(self.h, R.ptr(x), C.byref(s), R.ptr(first)))
        return x, s, first

    def window_set_allreduce(self, fn):
        """fn(d_ptr: int, count: int) -> None; kept alive on the context"""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)

        def _tramp(user, ptr, count):
            try:
                fn(ptr, count)
                return 0
            except Exception as e:  # pragma: no cover
                print("allreduce callback failed:", e)
                return 1

        self._cb = CB(_tramp) if fn else C.cast(None, CB)
        self._ck(self.lib.wc_window_set_allreduce(self.h, self._cb, C.c_void_p(0)))

    # ---- correspondence ---------------------------------------------------------------------------------------------
    def match_device(self, d_q_surf, d_q_pose, nq, d_t_surf, d_t_pose, nt, same_set, d_pairs, cap, d_knn_idx=None, d_knn_d2=None, sharded=False):
        """sharded: wc_match_sharded - a collective of the ctx's communicator (every rank, same replicated arguments)"""
        n = C.c_uint64(0)
        if sharded:
            self._ck(self.lib.wc_match_sharded(self.h, C.c_void_p(d_q_surf.ptr), C.c_void_p(d_q_pose.ptr), C.c_uint64(nq), C.c_void_p(d_t_surf.ptr),
                                               C.c_void_p(d_t_pose.ptr), C.c_uint64(nt), C.c_int(1 if same_set else 0), C.c_void_p(d_pairs.ptr),
                                               C.c_uint64(cap), C.byref(n)))
            return int(n.value)
        self._ck(self.lib.wc_match(self.h, C.c_void_p(d_q_surf.ptr), C.c_void_p(d_q_pose.ptr), C.c_uint64(nq), C.c_void_p(d_t_surf.ptr),
                                   C.c_void_p(d_t_pose.ptr), C.c_uint64(nt), C.c_int(1 if same_set else 0), C.c_void_p(d_pairs.ptr), C.c_uint64(cap),
                                   C.byref(n), C.c_void_p(d_knn_idx.ptr if d_knn_idx else 0), C.c_void_p(d_knn_d2.ptr if d_knn_d2 else 0)))
        return int(n.value)

    def match_stats(self):
        """what the last wc_match on this context touched, per sampled query (wc_match_stats)"""
        out = (C.c_double * 8)()
        self._ck(self.lib.wc_match_stats(self.h, out))
        n = max(out[4], 1.0)
        return {"nodes_per_query": out[0] / n, "leaves_per_query": out[1] / n, "points_per_query": out[2] / n, "exact_per_query": out[3] / n,
                "sampled_queries": int(out[4]), "tree_depth": int(out[5]), "top_levels": int(out[6]), "targets": int(out[7])}

    def match_pair_device(self, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, d_pairs_sld, cap_sld, d_pairs_fix, cap_fix, sharded=False):
        """both searches of an outer iteration side by side (wc_match_pair; sharded: wc_match_pair_sharded, a collective)
        -> (n_pairs_sld, n_pairs_fix)"""
        nb, nu = C.c_uint64(0), C.c_uint64(0)
        fn = self.lib.wc_match_pair_sharded if sharded else self.lib.wc_match_pair
        self._ck(fn(self.h, C.c_void_p(d_sld_surf.ptr), C.c_void_p(d_sld_pose.ptr), C.c_uint64(n_sld), C.c_void_p(d_fix_surf.ptr),
                                        C.c_void_p(d_fix_pose.ptr), C.c_uint64(n_fix), C.c_void_p(d_pairs_sld.ptr), C.c_uint64(cap_sld), C.byref(nb),
                                        C.c_void_p(d_pairs_fix.ptr), C.c_uint64(cap_fix), C.byref(nu)))
        return int(nb.value), int(nu.value)

    def match(self, q_surf, q_pose, t_surf, t_pose, same_set, want_knn=False, sharded=False):
        """host convenience -> pairs[PAIR] (and the raw k-NN table if want_knn)"""
        nq, nt = len(q_surf), len(t_surf)
        dq, dqp = self.to_device(q_surf), self.to_device(q_pose)
        if same_set:
            dt, dtp = dq, dqp
        else:
            dt, dtp = self.to_device(t_surf), self.to_device(t_pose)
        d_pairs = self.alloc(8 * max(nq, 1))
        k = self.params.knn_k
        d_idx = self.alloc(4 * max(nq, 1) * k) if want_knn else None
        d_d2 = self.alloc(8 * max(nq, 1) * k) if want_knn else None
        n = self.match_device(dq, dqp, nq, dt, dtp, nt, same_set, d_pairs, nq, d_idx, d_d2, sharded=sharded)
        pairs = d_pairs.download(R.PAIR, n)
        if want_knn:
            return pairs, d_idx.download(np.uint32, nq * k).reshape(nq, k), d_d2.download(np.float64, nq * k).reshape(nq, k)
        return pairs


class Odometry:
    """the C++ LidarOdometry facade (host/lidar_odometry.h) through its flat C wrapper (host/odom_c_api.cc)"""

    def __init__(self, device=0):
        so = os.path.join(os.path.dirname(_CSRC), "host", "libwildcat_odometry.so")
        load()
        self.lib = C.CDLL(so)
        self.lib.wc_odom_create.restype = C.c_void_p
        self.lib.wc_odom_num_samples.restype = C.c_uint64
        self.lib.wc_odom_append_ms.restype = C.c_double
        self.h = C.c_void_p(self.lib.wc_odom_create(C.c_int(device)))

    def close(self):
        if self.h:
            self.lib.wc_odom_destroy(self.h)
            self.h = None

    def add_imu(self, t, acc, gyr):
        a, g = (C.c_double * 3)(*acc), (C.c_double * 3)(*gyr)
        self.lib.wc_odom_add_imu(self.h, C.c_double(t), a, g)

    def add_scan(self, points):
        assert points.dtype == R.POINT
        self.lib.wc_odom_add_scan(self.h, R.ptr(points), C.c_uint64(len(points)))

    def stage_ms(self):
        """wall time [ms] of the last completed sweep's stages + its LM iterations (host/lidar_odometry.cc)"""
        out = (C.c_double * 8)()
        self.lib.wc_odom_stage_ms(self.h, out)
        names = ("predict_undistort", "extract_poses", "match", "build", "solve", "update", "shrink")
        d = dict(zip(names, [float(v) for v in out[:7]]))
        d["lm_iterations"] = int(out[7])
        d["append"] = float(self.lib.wc_odom_append_ms(self.h))  # upload + pre-filter of the completing message, in front of the stages
        return d

    def extract_paths(self):
        """(sweeps extracted by the default integer-moment path, sweeps extracted in the reference's summation order)"""
        out = (C.c_int * 2)()
        self.lib.wc_odom_extract_paths(self.h, out)
        return out[0], out[1]

    def set_fill_outputs(self, on):
        """what the reference publishes after a sweep (lidar_odometry.cc:582-602) as plain data: markers, sweep, tf"""
        self.lib.wc_odom_set_fill_outputs(self.h, C.c_int(1 if on else 0))

    def outputs(self):
        """-> dict(markers (n, 14): position, orientation wxyz, scale, rgba; scan: POINT records of the published sweep; stamp;
        tf: stamp, origin (3), rotation xyzw (4))"""
        self.lib.wc_odom_markers.restype = C.c_uint64
        self.lib.wc_odom_scan_in_world.restype = C.c_uint64
        n = int(self.lib.wc_odom_markers(self.h, None, C.c_uint64(0)))
        m = np.zeros((max(n, 1), 14))
        self.lib.wc_odom_markers(self.h, R.ptr(m), C.c_uint64(n))
        k = int(self.lib.wc_odom_scan_in_world(self.h, None, C.c_uint64(0), None, None))
        pts = np.zeros(max(k, 1), R.POINT)
        stamp, tf = C.c_double(0), np.zeros(8)
        self.lib.wc_odom_scan_in_world(self.h, R.ptr(pts), C.c_uint64(k), C.byref(stamp), R.ptr(tf))
        return dict(markers=m[:n], scan=pts[:k], stamp=stamp.value, tf=tf)

    def set_residual_log(self, on):
        """the reference's residual histograms before / after every solve (lidar_odometry.cc:56-94, :547-549, :568-570)"""
        self.lib.wc_odom_set_residual_log(self.h, C.c_int(1 if on else 0))

    def residual_log(self):
        self.lib.wc_odom_residual_log.restype = C.c_uint64
        n = int(self.lib.wc_odom_residual_log(self.h, None, C.c_uint64(0)))
        buf = C.create_string_buffer(n + 1)
        self.lib.wc_odom_residual_log(self.h, buf, C.c_uint64(n + 1))
        return buf.value.decode()

    def set_dev_option(self, name, value):
        rc = self.lib.wc_odom_set_dev_option(self.h, name.encode(), C.c_int(int(value)))
        if rc != 0:
            raise WildcatError(rc, "wc_odom_set_dev_option(%s)" % name)

    def set_quirks(self, on):
        self.lib.wc_odom_set_quirks(self.h, C.c_int(1 if on else 0))

    def set_exact_sums(self, on):
        self.lib.wc_odom_set_exact_sums(self.h, C.c_int(1 if on else 0))

    def import_state(self, samples23, imu):
        """test hook: continue from another run's sample / IMU states (LidarOdometry::ImportState)"""
        samples23 = np.ascontiguousarray(samples23, dtype=np.float64)
        imu = np.ascontiguousarray(imu)
        assert samples23.shape[1] == 23 and imu.dtype == R.IMU_STATE
        rc = self.lib.wc_odom_import_state(self.h, R.ptr(samples23), C.c_uint64(len(samples23)), R.ptr(imu), C.c_uint64(len(imu)))
        assert rc == 0, "import_state: the window's sample / IMU state counts differ"

    def sweeps(self):
        return int(self.lib.wc_odom_sweeps(self.h))

    def samples(self):
        n = int(self.lib.wc_odom_num_samples(self.h))
        out = np.zeros((n, 15))
        for i in range(n):
            self.lib.wc_odom_sample(self.h, C.c_uint64(i), R.ptr(out[i]))
        return out

    def set_keep_pair_stamps(self, on):
        self.lib.wc_odom_set_keep_pair_stamps(self.h, C.c_int(1 if on else 0))

    def pair_stamps(self, which):
        """test hook: (first, second) surfel timestamps of the last sweep's correspondences (0: sliding, 1: fixed window) -> float64[n, 2]"""
        self.lib.wc_odom_pair_stamps.restype = C.c_uint64
        n = int(self.lib.wc_odom_pair_stamps(self.h, C.c_int(which), None, C.c_uint64(0)))
        out = np.zeros(max(n, 2))
        self.lib.wc_odom_pair_stamps(self.h, C.c_int(which), R.ptr(out), C.c_uint64(n))
        return out[:n].reshape(-1, 2)

    def fixed_times(self):
        self.lib.wc_odom_fixed_times.restype = C.c_uint64
        n = int(self.lib.wc_odom_fixed_times(self.h, None, C.c_uint64(0)))
        out = np.zeros(max(n, 1))
        self.lib.wc_odom_fixed_times(self.h, R.ptr(out), C.c_uint64(n))
        return out[:n]

    def stats(self):
        s = np.zeros(8)
        self.lib.wc_odom_stats(self.h, R.ptr(s))
        return dict(zip(("sld_surfels", "fix_surfels", "binary", "unary", "lm_iters", "cost0", "cost1", "termination"), s.tolist()))
