"""Multi-GPU plumbing of the window solve: correspondences and IMU factors are sharded over the ranks,
the spline state stays replicated, and ONE sum-all-reduce of the packed normal equations per linearisation - the upper
block triangle of H in block-pair order (144 doubles per pair of sample blocks), g (np) and the cost - plus one scalar
per candidate-cost evaluation keeps every rank's LM state identical.  torch.distributed is plumbing: backend "nccl" is
RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous, balanced slice [lo, lo + count) of n items for `rank`"""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi - lo


def shard_imu(imu, rank, world):
    """IMU factors are triples of consecutive states (BuildImuResiduals, lidar_odometry.cc:319-363): factor i = states
    i, i+1, i+2.  A rank gets a contiguous share of the factors, i.e. its states plus the two that follow."""
    nf = max(0, len(imu) - 2)
    lo, cnt = shard_range(nf, rank, world)
    return imu[lo : lo + cnt + 2] if cnt > 0 else imu[:0]


def _padded(n):
    return ((n + 1 + 31) // 32) * 32


def pair_offsets(ns):
    """offset of every upper block pair (I <= J, pair order) in the reduction buffer, and the buffer's H part in doubles: 144
    doubles for a pair of sample blocks at most two apart (IMU factors reach that far, cost_functor.h:264-355), the 6 x 6 pose
    corner (36) for the others, which only surfel factors touch (csrc/window.hip: wc_window_state::pair_off)"""
    off, o = {}, 0
    for i in range(ns):
        for j in range(i, ns):
            off[(i, j)] = o
            o += 144 if j - i <= 2 else 36
    return off, o


def packed_count(ns):
    """number of doubles in the reduction buffer {upper block pairs, g, cost, spare} of a window with ns sample states"""
    return pair_offsets(ns)[1] + _padded(12 * ns) + 2


def corner_count(ns):
    """doubles the ranks of a sharded window sum per linearisation in the two-collective form (round 6, csrc/window.hip: k_gather only = 1):
    the 6 x 6 pose corner of EVERY upper block pair (36 npairs), the pose half of g (6 ns) - the large collective - and {cost of the
    surfel factors, spare} - the 16-byte collective that goes first"""
    return 36 * (ns * (ns + 1) // 2) + 6 * ns + 2


def pack_corners(H, g):
    """the surfel factors' H, g of ONE rank (their bias rows and columns are zero) -> the large collective's buffer"""
    n = len(g)
    ns = n // 12
    buf = np.zeros(corner_count(ns) - 2)
    pid = 0
    for i in range(ns):
        for j in range(i, ns):
            blk = H[12 * i : 12 * i + 12, 12 * j : 12 * j + 12]
            assert not blk[6:, :].any() and not blk[:, 6:].any(), "a surfel factor touched a bias unknown"
            buf[36 * pid : 36 * pid + 36] = blk[:6, :6].reshape(-1)
            pid += 1
    gg = np.asarray(g).reshape(ns, 12)
    assert not gg[:, 6:].any()
    buf[36 * pid :] = gg[:, :6].reshape(-1)
    return buf


def add_corners(buf, H_imu, g_imu):
    """the summed buffer on top of the (replicated) IMU factors' H, g -> the window's H, g (what k_expand_corners does on the device)"""
    n = len(g_imu)
    ns = n // 12
    H, g = np.array(H_imu, copy=True), np.array(g_imu, copy=True)
    pid = 0
    for i in range(ns):
        for j in range(i, ns):
            c = buf[36 * pid : 36 * pid + 36].reshape(6, 6)
            H[12 * i : 12 * i + 6, 12 * j : 12 * j + 6] += c
            if i != j:
                H[12 * j : 12 * j + 6, 12 * i : 12 * i + 6] += c.T
            pid += 1
    g.reshape(ns, 12)[:, :6] += buf[36 * pid :].reshape(ns, 6)
    return H, g


def pack(H, g, cost):
    """dense H (n x n), g, cost -> the reduction buffer (the layout k_gather writes for a sharded problem)"""
    n = len(g)
    ns = n // 12
    off, nh = pair_offsets(ns)
    buf = np.zeros(nh + _padded(n) + 2)
    for (i, j), o in off.items():
        blk = H[12 * i : 12 * i + 12, 12 * j : 12 * j + 12]
        if j - i <= 2:
            buf[o : o + 144] = blk.reshape(-1)
        else:
            assert not blk[6:, :].any() and not blk[:, 6:].any(), "a far pair has entries outside its pose corner"
            buf[o : o + 36] = blk[:6, :6].reshape(-1)
    buf[nh : nh + n] = g
    buf[nh + _padded(n)] = cost
    return buf


def unpack(buf, ns):
    """the reduction buffer -> dense symmetric H, g, cost (what k_expand_pairs does on the device)"""
    n = 12 * ns
    off, nh = pair_offsets(ns)
    H = np.zeros((n, n))
    for (i, j), o in off.items():
        blk = np.zeros((12, 12))
        if j - i <= 2:
            blk[:] = buf[o : o + 144].reshape(12, 12)
        else:
            blk[:6, :6] = buf[o : o + 36].reshape(6, 6)
        H[12 * i : 12 * i + 12, 12 * j : 12 * j + 12] = blk
        H[12 * j : 12 * j + 12, 12 * i : 12 * i + 12] = blk.T
    return H, buf[nh : nh + n], float(buf[nh + _padded(n)])


class DeviceView:
    """zero-copy torch view of a raw device pointer of doubles (through __cuda_array_interface__)"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def make_allreduce(torch, dist, device):
    """callback for Context.window_set_allreduce: sums a device buffer over all ranks with RCCL"""

    def allreduce(ptr, count):
        t = torch.as_tensor(DeviceView(ptr, count), device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)

    return allreduce


# ---- communicators for Context.set_comm (the wc_comm callbacks of include/wc_types.h) --------------------------------------
class ThreadComm:
    """`world` contexts inside ONE process (one per thread) exchange through host staging - the 1-GPU stand-in for RCCL used
    by the GPU tests.  shared = ThreadComm.shared(world); rank r's context gets ThreadComm(shared, r, ctx)."""

    @staticmethod
    def shared(world):
        import threading

        return {"world": world, "bar": threading.Barrier(world), "slot": [None] * world, "calls": [0] * world}

    def __init__(self, shared, rank, ctx):
        self.s, self.rank, self.world, self.ctx = shared, rank, shared["world"], ctx

    def _publish(self, obj):
        self.s["slot"][self.rank] = obj
        self.s["bar"].wait()
        got = list(self.s["slot"])
        self.s["bar"].wait()
        self.s["calls"][self.rank] += 1
        return got

    def allreduce(self, ptr, count):
        mine = self.ctx.download_raw(ptr, count * 8).view(np.float64).copy()
        total = np.zeros_like(mine)
        for part in self._publish(mine):  # fixed rank order: bitwise the same sum on every rank
            total = total + part
        self.ctx.upload_raw(ptr, total)

    def alltoallv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        total = int(sum(send_bytes))
        mine = self.ctx.download_raw(send_ptr, total) if total else np.zeros(0, np.uint8)
        offs = np.concatenate([[0], np.cumsum(send_bytes)]).astype(np.int64)
        parts = [mine[offs[r] : offs[r + 1]].copy() for r in range(self.world)]
        got = self._publish(parts)
        recv = [got[src][self.rank] for src in range(self.world)]
        assert [len(p) for p in recv] == [int(b) for b in recv_bytes], "alltoallv: receive counts disagree"
        buf = np.concatenate(recv) if recv else np.zeros(0, np.uint8)
        if len(buf):
            self.ctx.upload_raw(recv_ptr, buf)

    def allgatherv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        mine = self.ctx.download_raw(send_ptr, int(send_bytes)) if send_bytes else np.zeros(0, np.uint8)
        got = self._publish(mine.copy())
        assert [len(p) for p in got] == [int(b) for b in recv_bytes], "allgatherv: receive counts disagree"
        buf = np.concatenate(got)
        if len(buf):
            self.ctx.upload_raw(recv_ptr, buf)


class ByteView:
    """zero-copy torch view of `nbytes` raw device bytes"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """the same three collectives through torch.distributed on the device buffers themselves (backend "nccl" = RCCL over xGMI)"""

    def __init__(self, torch, dist, device):
        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

        self.on_gpu = str(device) != "cpu"  # "cpu": the pointers are host addresses (gloo, CPU tests)

    def _bytes(self, ptr, n):
        if n == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device=self.device)
        if not self.on_gpu:
            import ctypes

            return self.torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * int(n)).from_address(int(ptr))))
        return self.torch.as_tensor(ByteView(ptr, n), device=self.device)

    def _done(self):
        if self.on_gpu:
            self.torch.cuda.synchronize(self.device)

    def allreduce(self, ptr, count):
        t = self._bytes(ptr, 8 * count).view(self.torch.float64)
        self.dist.all_reduce(t)
        self._done()

    def alltoallv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        s, r = self._bytes(send_ptr, int(sum(send_bytes))), self._bytes(recv_ptr, int(sum(recv_bytes)))
        self.dist.all_to_all_single(r, s, [int(b) for b in recv_bytes], [int(b) for b in send_bytes])
        self._done()

    def allgatherv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        # contributions differ in size: an all-to-all in which every rank sends its whole contribution to every rank
        s = self._bytes(send_ptr, int(send_bytes)).repeat(self.world)
        r = self._bytes(recv_ptr, int(sum(recv_bytes)))
        self.dist.all_to_all_single(r, s, [int(b) for b in recv_bytes], [int(send_bytes)] * self.world)
        self._done()


class StagedTorchComm:
    """the three collectives through torch.distributed on HOST copies of the device buffers (wc_d2h / wc_h2d of the context that
    owns them): lets several PROCESSES share one GPU under the gloo backend - RCCL refuses two ranks on one device - so that the
    process-level plumbing of the N-rank step (one context per process, no shared Python state, rendezvous through
    torch.distributed) can be tested on a one-GPU box (tests/test_step_gpu.py)"""

    def __init__(self, torch, dist, ctx):
        import ctypes

        self.torch, self.dist, self.ctx, self.C = torch, dist, ctx, ctypes
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def _down(self, ptr, n):
        a = np.empty(max(int(n), 1), np.uint8)
        if n:
            self.ctx._ck(self.ctx.lib.wc_d2h(self.ctx.h, self.C.c_void_p(a.ctypes.data), self.C.c_void_p(int(ptr)), self.C.c_size_t(int(n))))
        return self.torch.from_numpy(a[: int(n)])

    def _up(self, ptr, t):
        a = np.ascontiguousarray(t.numpy())
        if a.nbytes:
            self.ctx._ck(self.ctx.lib.wc_h2d(self.ctx.h, self.C.c_void_p(int(ptr)), self.C.c_void_p(a.ctypes.data), self.C.c_size_t(a.nbytes)))

    def allreduce(self, ptr, count):
        t = self._down(ptr, 8 * count).view(self.torch.float64)
        self.dist.all_reduce(t)
        self._up(ptr, t.view(self.torch.uint8))

    def alltoallv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        s = self._down(send_ptr, int(sum(send_bytes)))
        r = self.torch.empty(int(sum(recv_bytes)), dtype=self.torch.uint8)
        self.dist.all_to_all_single(r, s, [int(b) for b in recv_bytes], [int(b) for b in send_bytes])
        self._up(recv_ptr, r)

    def allgatherv(self, send_ptr, send_bytes, recv_ptr, recv_bytes):
        s = self._down(send_ptr, int(send_bytes)).repeat(self.world)
        r = self.torch.empty(int(sum(recv_bytes)), dtype=self.torch.uint8)
        self.dist.all_to_all_single(r, s, [int(b) for b in recv_bytes], [int(send_bytes)] * self.world)
        self._up(recv_ptr, r)


def route_partition_host(points, keys_xyz, world):
    """host restatement of wc_route_partition (CPU / gloo test): stable partition of a POINT array by the owner of each
    point's root voxel -> list of `world` arrays, time order preserved inside each"""
    from . import lib

    owners = lib.route_owner(keys_xyz, world)
    return [points[owners == r] for r in range(world)]


def merge_surfels_host(lists, id_lists):
    """host restatement of wc_merge_surfels: the canonical order (timestamp, kx, ky, kz, node) over the concatenation"""
    s, i = np.concatenate(lists), np.concatenate(id_lists)
    order = np.lexsort((i["node"], i["kz"], i["ky"], i["kx"], s["t"]))
    return s[order], i[order]
