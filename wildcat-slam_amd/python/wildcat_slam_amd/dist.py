"""Multi-GPU plumbing of the window solve: correspondences are sharded over the ranks, the spline
state stays replicated, and ONE sum-all-reduce of the packed normal equations {H (n*n), g (np), cost} per linearisation
(plus one scalar per candidate-cost evaluation) keeps every rank's LM state identical.  torch.distributed is plumbing:
backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous, balanced slice [lo, lo + count) of n items for `rank`"""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi - lo


def packed_count(ns):
    """number of doubles in the packed {H, g, cost, spare} buffer of a window with ns sample states"""
    n = 12 * ns
    np_ = ((n + 1 + 31) // 32) * 32
    return n * n + np_ + 2


def pack(H, g, cost):
    n = len(g)
    np_ = ((n + 1 + 31) // 32) * 32
    buf = np.zeros(n * n + np_ + 2)
    buf[: n * n] = H.reshape(-1)
    buf[n * n : n * n + n] = g
    buf[n * n + np_] = cost
    return buf


def unpack(buf, ns):
    n = 12 * ns
    np_ = ((n + 1 + 31) // 32) * 32
    return buf[: n * n].reshape(n, n), buf[n * n : n * n + n], float(buf[n * n + np_])


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
