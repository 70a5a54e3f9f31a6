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


def packed_count(ns):
    """number of doubles in the reduction buffer {upper block pairs, g, cost, spare} of a window with ns sample states"""
    return (ns * (ns + 1) // 2) * 144 + _padded(12 * ns) + 2


def pack(H, g, cost):
    """dense H (n x n), g, cost -> the reduction buffer (the layout k_gather writes when an all-reduce is installed)"""
    n = len(g)
    ns = n // 12
    npairs = ns * (ns + 1) // 2
    buf = np.zeros(npairs * 144 + _padded(n) + 2)
    pid = 0
    for i in range(ns):
        for j in range(i, ns):
            buf[pid * 144 : (pid + 1) * 144] = H[12 * i : 12 * i + 12, 12 * j : 12 * j + 12].reshape(-1)
            pid += 1
    buf[npairs * 144 : npairs * 144 + n] = g
    buf[npairs * 144 + _padded(n)] = cost
    return buf


def unpack(buf, ns):
    """the reduction buffer -> dense symmetric H, g, cost (what k_expand_pairs does on the device)"""
    n = 12 * ns
    npairs = ns * (ns + 1) // 2
    H = np.zeros((n, n))
    pid = 0
    for i in range(ns):
        for j in range(i, ns):
            blk = buf[pid * 144 : (pid + 1) * 144].reshape(12, 12)
            H[12 * i : 12 * i + 12, 12 * j : 12 * j + 12] = blk
            H[12 * j : 12 * j + 12, 12 * i : 12 * i + 12] = blk.T
            pid += 1
    return H, buf[npairs * 144 : npairs * 144 + n], float(buf[npairs * 144 + _padded(n)])


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
