"""One rank of the N-rank odometry step as a PROCESS of its own (tests/test_step_gpu.py starts two of them on the one GPU of the
box): torch.distributed over gloo for the rendezvous and the collectives (dist.StagedTorchComm), one wc_ctx per process.  Writes the
rank's result to <out>.rank<r>.npz.  Usage: python _step_worker.py <out-prefix> <scans> <roots>; RANK / WORLD_SIZE / MASTER_* from the
environment."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wildcat-slam_amd", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from wildcat_slam_amd import dist as wdist, lib, synth  # noqa: E402
from wildcat_slam_amd.step import StepWindow  # noqa: E402


def main():
    out, scans, roots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = lib.Context(0)
    ctx.set_comm(wdist.StagedTorchComm(torch, dist, ctx))
    w = synth.g2_scan_sequence(scans, roots, m=32, seed=synth.SEED + 5)
    sw = StepWindow(ctx, w, rank=rank, world=world)
    _, info, x = sw.step()
    np.savez(out + ".rank%d.npz" % rank, x=x, **{k: np.asarray(v) for k, v in info.items()})
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
