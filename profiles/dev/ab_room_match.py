# the two-set search on ROOM surfels (sweeps of a spinning scanner in firing order: surfels in time-bin order), development options as in ab_match_opt.py
import os, sys, time, zlib
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth, records as R
ctx = lib.Context(0)
DEF = {"knn_sort": -1, "knn_early": 1, "knn_group": -1}
for pps in (640_000, 1_600_000):
    msgs, _, _ = synth.raw_stream(4.0, pts_per_s=pps, t_start=1000.0)
    surf = []
    for k in range(0, len(msgs) - 4, 5):
        s, _ = ctx.extract_surfels(synth.concat_points(*msgs[k:k + 5]))
        surf.append(s)
    F, S = np.concatenate(surf[:2]), np.concatenate(surf[2:])
    PF, PS = np.zeros(len(F), R.POSE), np.zeros(len(S), R.POSE)
    PF["quat"][:, 0] = 1.0; PS["quat"][:, 0] = 1.0
    d_s, d_p, d_fs, d_fp = ctx.to_device(S), ctx.to_device(PS), ctx.to_device(F), ctx.to_device(PF)
    n_s, n_f = len(S), len(F)
    d_b, d_u = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    for spec in (sys.argv[1:] or [""]):
        kv = [s.split("=") for s in spec.split(",") if s]
        for k, v in kv: ctx.set_dev_option(k, int(v))
        res = []
        for which in ("same", "fixed", "pair"):
            ts = []
            for rep in range(7):
                ctx.sync(); t0 = time.perf_counter()
                if which == "same": n = ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
                elif which == "fixed": n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
                else: n = ctx.match_pair_device(d_s, d_p, n_s, d_fs, d_fp, n_f, d_b, n_s, d_u, n_s)
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[1:])
            res.append("%s %.3f" % (which, ts[len(ts) // 2] * 1e3))
        crc = zlib.crc32(d_u.download(np.uint8, 8 * int(n[1])).tobytes(), zlib.crc32(d_b.download(np.uint8, 8 * int(n[0])).tobytes()))
        for k, v in kv: ctx.set_dev_option(k, DEF.get(k, 0))
        print("room %d pts/s: %d queries, %d fixed [%-22s] %s ms  pairs %s crc %08x" % (pps, n_s, n_f, spec, "  ".join(res), n, crc), flush=True)
