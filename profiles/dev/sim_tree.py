#!/usr/bin/env python3
"""CPU simulation (numpy, no GPU) of candidate 6-D indices for the matcher: counts what a query would touch - internal
nodes whose child boxes are loaded, leaves whose points get a first look, points - for an implicit B-ary bounding-box
tree over the targets in space-filling-curve order.  Design aid for csrc/match.hip; uses the CPU
oracle for the extraction of the workloads' surfels (this is a profiling script, not product code)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from wildcat_slam_amd import records as R, synth  # noqa: E402

AS = 5.0 * np.pi / 180.0


def quat_rot(q, v):
    w, x, y, z = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:4]
    u = np.concatenate([x, y, z], 1)
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def features(surf, pose):
    cw = quat_rot(pose["quat"], surf["center"]) + pose["pos"]
    nw = quat_rot(pose["quat"], surf["normal"])
    return np.concatenate([cw, nw / AS], 1)


def morton_keys(F, bits, dims=6, cell=None):
    lo = F.min(0)
    ext = (F.max(0) - lo).max()
    if cell is None:
        cell = ext / (1 << bits) * (1 + 1e-9)
    q = np.minimum(((F - lo) / cell).astype(np.int64), (1 << bits) - 1)
    key = np.zeros(len(F), np.uint64)
    for b in range(bits):
        for d in range(dims):
            key |= ((q[:, d] >> b) & 1).astype(np.uint64) << np.uint64(b * dims + d)
    return key


def octa(n):
    n = n / np.abs(n).sum(1, keepdims=True)
    u, v = n[:, 0].copy(), n[:, 1].copy()
    neg = n[:, 2] < 0
    uu = (1 - np.abs(v)) * np.sign(u + 1e-300)
    vv = (1 - np.abs(u)) * np.sign(v + 1e-300)
    u[neg], v[neg] = uu[neg], vv[neg]
    return np.stack([u, v], 1)


def keys_5d(F, bits):
    """3 spatial dims + octahedral (u, v) of the normal scaled to the sphere's size in units"""
    uv = octa(F[:, 3:6] * AS) * (np.pi / 2 / AS)  # ~ arc length units
    G = np.concatenate([F[:, :3], uv], 1)
    return morton_keys(G, bits, dims=5)


class Tree:
    def __init__(self, F, order, L, B):
        self.F = F[order]
        self.order = order
        self.L, self.B = L, B
        n = len(F)
        nl = (n + L - 1) // L
        pad = nl * L - n
        Fp = np.concatenate([self.F, np.repeat(self.F[-1:], pad, 0)]) if pad else self.F
        lo = Fp.reshape(nl, L, 6).min(1)
        hi = Fp.reshape(nl, L, 6).max(1)
        self.lo, self.hi = [lo], [hi]
        while len(lo) > 1:
            m = (len(lo) + B - 1) // B
            pad = m * B - len(lo)
            if pad:
                lo = np.concatenate([lo, np.repeat(lo[-1:], pad, 0)])
                hi = np.concatenate([hi, np.repeat(hi[-1:], pad, 0)])
            lo = lo.reshape(m, B, 6).min(1)
            hi = hi.reshape(m, B, 6).max(1)
            self.lo.append(lo)
            self.hi.append(hi)
        self.levels = len(self.lo)

    def box_d2(self, lvl, i0, i1, q):
        lo, hi = self.lo[lvl][i0:i1], self.hi[lvl][i0:i1]
        d = np.maximum(np.maximum(lo - q, q - hi), 0.0)
        return (d * d).sum(1)

    def query(self, q, k, seed_pos=None, near_first=True):
        """-> (kth d2, internal visits, leaf visits, points, exact evals)"""
        L, B = self.L, self.B
        best = np.full(k, np.inf)
        n = len(self.F)
        st = dict(nodes=0, leaves=0, pts=0)
        done_leaf = -1

        def scan_leaf(li):
            nonlocal best
            a, b = li * L, min(n, (li + 1) * L)
            d = ((self.F[a:b] - q) ** 2).sum(1)
            best = np.sort(np.concatenate([best, d]))[:k]
            st["leaves"] += 1
            st["pts"] += b - a

        if seed_pos is not None:
            done_leaf = seed_pos // L
            scan_leaf(done_leaf)
        top = self.levels - 1
        stack = [(top, 0, 0.0)]
        while stack:
            lvl, i, dd = stack.pop()
            if dd >= best[-1]:
                continue
            if lvl == 0:
                if i != done_leaf:
                    scan_leaf(i)
                continue
            c0, c1 = i * B, min((i + 1) * B, len(self.lo[lvl - 1]))
            d2 = self.box_d2(lvl - 1, c0, c1, q)
            st["nodes"] += 1
            idx = np.arange(c0, c1)
            keep = d2 < best[-1]
            idx, d2 = idx[keep], d2[keep]
            if near_first:
                o = np.argsort(-d2, kind="stable")
                idx, d2 = idx[o], d2[o]
            else:
                idx, d2 = idx[::-1], d2[::-1]
            for j, dj in zip(idx, d2):
                stack.append((lvl - 1, int(j), float(dj)))
        return best[-1], st["nodes"], st["leaves"], st["pts"]


def run(name, Fq, Ft, same, L, B, keyfn, nsamp=400, k=10, seed=True):
    t0 = time.time()
    keys = keyfn(Ft)
    order = np.argsort(keys, kind="stable")
    T = Tree(Ft, order, L, B)
    rng = np.random.default_rng(1)
    if same:
        pos = rng.choice(len(Ft), nsamp, replace=False)
        qs = T.F[pos]
        seeds = pos
    else:
        sel = rng.choice(len(Fq), nsamp, replace=False)
        qs = Fq[sel]
        # seed: position of the query's key among the target keys (needs the same quantisation: recompute on the union)
        both = np.concatenate([Ft, qs])
        kb = keyfn(both)
        ks = np.sort(kb[: len(Ft)])
        # the order of Ft under the union's quantisation may differ slightly from `order`; good enough for a seed
        seeds = np.minimum(np.searchsorted(ks, kb[len(Ft):]), len(Ft) - 1)
    res = []
    for q, s in zip(qs, seeds):
        res.append(T.query(q, k, int(s) if seed else None))
    res = np.array(res)
    print("%-28s L=%2d B=%2d lv=%d: kth %.2f | nodes %.1f (p90 %.0f max %.0f)  leaves %.1f (p90 %.0f)  pts %.0f (p90 %.0f)  [%.1fs]" % (
        name, L, B, T.levels, np.sqrt(np.median(res[:, 0])), res[:, 1].mean(), np.percentile(res[:, 1], 90), res[:, 1].max(),
        res[:, 2].mean(), np.percentile(res[:, 2], 90), res[:, 3].mean(), np.percentile(res[:, 3], 90), time.time() - t0), flush=True)


def workloads(which):
    out = {}
    if "step" in which:
        roots = int(os.environ.get("ROOTS", "3906"))
        w = synth.g2_scan_sequence(10, roots, m=32, seed=synth.SEED + 21)
        surf = [pyoracle.extract_surfels(s)[0] for s in w["scans"]]
        S = np.concatenate(surf)
        P = np.zeros(len(S), R.POSE)
        Bf = np.zeros(len(S), np.uint8)
        pyoracle.update_surfel_poses(w["imu"], S, P, Bf)
        F = features(S, P)
        nfix = len(surf[0]) + len(surf[1])
        out["step_fix"] = (F[nfix:], F[:nfix], False)
        out["step_sld"] = (F[nfix:], F[nfix:], True)
    if "room" in which:
        msgs, _, _ = synth.raw_stream(4.0, pts_per_s=640_000, t_start=1000.0)
        surf = []
        for k in range(0, len(msgs) - 4, 5):
            surf.append(pyoracle.extract_surfels(synth.concat_points(*msgs[k:k + 5]))[0])
        S = np.concatenate(surf)
        P = np.zeros(len(S), R.POSE)
        P["quat"][:, 0] = 1.0
        F = features(S, P)
        out["room"] = (F, F, True)
    if "c4" in which:
        n = int(os.environ.get("C4P", "50000"))
        w = synth.surfel_window(20, n, seed=synth.SEED + 7, fixed_patches=n)
        F = features(w["surf"], w["pose"])
        Ff = features(w["fix_surf"], w["fix_pose"])
        out["c4_sld"] = (F, F, True)
        out["c4_fix"] = (F, Ff, False)
    return out


if __name__ == "__main__":
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["step", "room"]
    W = workloads(which)
    for name, (Fq, Ft, same) in W.items():
        print(name, "queries", len(Fq), "targets", len(Ft), "extent", np.round(Ft.max(0) - Ft.min(0), 1))
        for L, B in ((8, 8), (8, 4), (4, 8), (16, 8)):
            for kn, kf in (("m6x10", lambda F: morton_keys(F, 10)), ("m5x12", lambda F: keys_5d(F, 12))):
                run(name + " " + kn, Fq, Ft, same, L, B, kf)


# ---- a balanced kd-tree (median split on the widest dimension of the node's box), boxes per node: what an ideal build would give
class KdTree:
    def __init__(self, F, L):
        self.F = F
        self.L = L
        n = len(F)
        self.perm = np.arange(n)
        self.nodes = []  # (lo, hi, left, right, a, b)

        def build(a, b):
            P = self.F[self.perm[a:b]]
            lo, hi = P.min(0), P.max(0)
            me = len(self.nodes)
            self.nodes.append(None)
            if b - a <= L:
                self.nodes[me] = (lo, hi, -1, -1, a, b)
                return me
            d = int(np.argmax(hi - lo))
            m = (a + b) // 2
            sub = self.perm[a:b]
            o = np.argpartition(self.F[sub, d], m - a)
            self.perm[a:b] = sub[o]
            l = build(a, m)
            r = build(m, b)
            self.nodes[me] = (lo, hi, l, r, a, b)
            return me

        sys.setrecursionlimit(10000)
        build(0, n)

    def query(self, q, k):
        best = np.full(k, np.inf)
        st = [0, 0, 0]

        def bd(node):
            lo, hi = node[0], node[1]
            d = np.maximum(np.maximum(lo - q, q - hi), 0.0)
            return float((d * d).sum())

        stack = [(0, 0.0)]
        while stack:
            i, dd = stack.pop()
            if dd >= best[-1]:
                continue
            nd = self.nodes[i]
            if nd[2] < 0:
                d = ((self.F[self.perm[nd[4]:nd[5]]] - q) ** 2).sum(1)
                best = np.sort(np.concatenate([best, d]))[:k]
                st[1] += 1
                st[2] += nd[5] - nd[4]
                continue
            st[0] += 1
            dl, dr = bd(self.nodes[nd[2]]), bd(self.nodes[nd[3]])
            if dl < dr:
                stack.append((nd[3], dr))
                stack.append((nd[2], dl))
            else:
                stack.append((nd[2], dl))
                stack.append((nd[3], dr))
        return best[-1], st[0], st[1], st[2]


def run_kd(name, Fq, Ft, same, L, nsamp=400, k=10):
    t0 = time.time()
    T = KdTree(Ft, L)
    rng = np.random.default_rng(1)
    qs = Ft[rng.choice(len(Ft), nsamp, replace=False)] if same else Fq[rng.choice(len(Fq), nsamp, replace=False)]
    res = np.array([T.query(q, k) for q in qs])
    # ideal counts: targets inside the 6-D ball's bounding cylinder (|dc| <= r and |dn| <= r) and inside its bounding cube
    cyl, cube = [], []
    for q, r2 in zip(qs[:100], res[:100, 0]):
        r = np.sqrt(r2)
        d = np.abs(Ft - q)
        cube.append(int((d.max(1) <= r).sum()))
        cyl.append(int((((d[:, :3] ** 2).sum(1) <= r2) & ((d[:, 3:] ** 2).sum(1) <= r2)).sum()))
    print("%-28s kd L=%2d: kth %.2f | nodes %.1f (p90 %.0f max %.0f)  leaves %.1f (p90 %.0f)  pts %.0f (p90 %.0f); in cylinder %.0f, in cube %.0f [%.1fs]" % (
        name, L, np.sqrt(np.median(res[:, 0])), res[:, 1].mean(), np.percentile(res[:, 1], 90), res[:, 1].max(), res[:, 2].mean(),
        np.percentile(res[:, 2], 90), res[:, 3].mean(), np.percentile(res[:, 3], 90), np.mean(cyl), np.mean(cube), time.time() - t0), flush=True)


def kd_order_inplace(F, perm, a, b, L):
    """balanced kd ordering of perm[a:b] (median split on the widest dimension) down to segments of L"""
    if b - a <= L:
        return
    P = F[perm[a:b]]
    d = int(np.argmax(P.max(0) - P.min(0)))
    m = a + ((b - a) // 2 + L - 1) // L * L if (b - a) > 2 * L else a + L  # keep leaf boundaries on multiples of L
    m = min(max(m, a + L), b - 1)
    o = np.argsort(P[:, d], kind="stable")
    perm[a:b] = perm[a:b][o]
    kd_order_inplace(F, perm, a, m, L)
    kd_order_inplace(F, perm, m, b, L)


def chunked_order(F, keys, C, L):
    perm = np.argsort(keys, kind="stable")
    n = len(F)
    for a in range(0, n, C):
        kd_order_inplace(F, perm, a, min(n, a + C), L)
    return perm


def run_chunked(name, Fq, Ft, same, L, B, C, keyfn, nsamp=300, k=10):
    t0 = time.time()
    order = chunked_order(Ft, keyfn(Ft), C, L)
    T = Tree(Ft, order, L, B)
    rng = np.random.default_rng(1)
    if same:
        pos = rng.choice(len(Ft), nsamp, replace=False)
        qs, seeds = T.F[pos], pos
    else:
        qs, seeds = Fq[rng.choice(len(Fq), nsamp, replace=False)], [None] * nsamp
    res = np.array([T.query(q, k, None if s is None else int(s)) for q, s in zip(qs, seeds)])
    print("%-24s L=%2d B=%2d C=%4d lv=%d: nodes %.1f (p90 %.0f max %.0f)  leaves %.1f (p90 %.0f)  pts %.0f (p90 %.0f)  [%.1fs]" % (
        name, L, B, C, T.levels, res[:, 1].mean(), np.percentile(res[:, 1], 90), res[:, 1].max(),
        res[:, 2].mean(), np.percentile(res[:, 2], 90), res[:, 3].mean(), np.percentile(res[:, 3], 90), time.time() - t0), flush=True)


# ---- implicit BALANCED binary tree: node (d, i) covers [n i / 2^d, n (i + 1) / 2^d); wide nodes = `w` levels collapsed
class BalTree:
    def __init__(self, F, L, w, order_fn="kd", keys=None, kd_levels=None):
        n = len(F)
        D = 0
        while (n + (1 << D) - 1) >> D > L:
            D += 1
        self.D, self.w, self.n = D, w, n
        perm = np.argsort(keys, kind="stable") if keys is not None else np.arange(n)
        # kd ordering for the levels [top_kd, D): segments of level top_kd are sorted recursively
        top = 0 if kd_levels is None else max(0, D - kd_levels)
        self.rng_ = lambda d, i: ((n * i) >> d, (n * (i + 1)) >> d)

        def rec(d, i):
            a, b = self.rng_(d, i)
            if d >= D or b - a <= 1:
                return
            P = F[perm[a:b]]
            dim = int(np.argmax(P.max(0) - P.min(0)))
            perm[a:b] = perm[a:b][np.argsort(P[:, dim], kind="stable")]
            rec(d + 1, 2 * i)
            rec(d + 1, 2 * i + 1)

        for i in range(1 << top):
            rec(top, i)
        self.F = F[perm]
        self.perm = perm
        # boxes per level
        self.lo, self.hi = [None] * (D + 1), [None] * (D + 1)
        idx = np.arange(1 << D)
        a, b = (n * idx) >> D, (n * (idx + 1)) >> D
        lo = np.array([self.F[x:y].min(0) for x, y in zip(a, b)])
        hi = np.array([self.F[x:y].max(0) for x, y in zip(a, b)])
        self.lo[D], self.hi[D] = lo, hi
        for d in range(D - 1, -1, -1):
            lo = np.minimum(lo[0::2], lo[1::2])
            hi = np.maximum(hi[0::2], hi[1::2])
            self.lo[d], self.hi[d] = lo, hi

    def query(self, q, k, seed_pos=None):
        D, w, n = self.D, self.w, self.n
        best = np.full(k, np.inf)
        st = [0, 0, 0]
        self.maxstack = 0
        done = -1

        def leaf(i):
            nonlocal best
            a, b = (n * i) >> D, (n * (i + 1)) >> D
            d = ((self.F[a:b] - q) ** 2).sum(1)
            best = np.sort(np.concatenate([best, d]))[:k]
            st[1] += 1
            st[2] += b - a

        if seed_pos is not None:
            # leaf of a position: largest i with (n i) >> D <= pos
            i = min((seed_pos << D) // n, (1 << D) - 1)
            while (n * (i + 1)) >> D <= seed_pos:
                i += 1
            while (n * i) >> D > seed_pos:
                i -= 1
            done = i
            leaf(i)
        if seed_pos is None and getattr(self, "greedy", True):
            d, i = 0, 0
            while d < D:  # greedy descent to the nearest leaf, nothing pushed
                dn = min(D, d + w)
                c0, c1 = i << (dn - d), (i + 1) << (dn - d)
                lo, hi = self.lo[dn][c0:c1], self.hi[dn][c0:c1]
                dv = np.maximum(np.maximum(lo - q, q - hi), 0.0)
                st[0] += 1
                if dn == D:
                    done = set(range(c0, c1))
                    for j in range(c0, c1):
                        leaf(j)
                d, i = dn, c0 + int(np.argmin((dv * dv).sum(1)))
        stack = [(0, 0, 0.0)]
        while stack:
            d, i, dd = stack.pop()
            if dd >= best[-1]:
                continue
            if d == D:
                if (i not in done) if isinstance(done, set) else (i != done):
                    leaf(i)
                continue
            dn = min(D, d + w)
            c0, c1 = i << (dn - d), (i + 1) << (dn - d)
            lo, hi = self.lo[dn][c0:c1], self.hi[dn][c0:c1]
            dv = np.maximum(np.maximum(lo - q, q - hi), 0.0)
            d2 = (dv * dv).sum(1)
            st[0] += 1
            o = np.argsort(-d2, kind="stable")
            for j in o:
                if d2[j] < best[-1]:
                    stack.append((dn, c0 + int(j), float(d2[j])))
            self.maxstack = max(self.maxstack, len(stack))
        return best[-1], st[0], st[1], st[2], self.maxstack


def run_bal(name, Fq, Ft, same, L, w, nsamp=300, k=10, keys=None, kd_levels=None, seed=True):
    t0 = time.time()
    T = BalTree(Ft, L, w, keys=keys, kd_levels=kd_levels)
    rng = np.random.default_rng(1)
    if same:
        pos = rng.choice(len(Ft), nsamp, replace=False)
        qs, seeds = T.F[pos], pos if seed else [None] * nsamp
    else:
        qs, seeds = Fq[rng.choice(len(Fq), nsamp, replace=False)], [None] * nsamp
    res = np.array([T.query(q, k, None if s is None else int(s)) for q, s in zip(qs, seeds)])
    print("%-24s L=%2d w=%d D=%2d kdlv=%s: kth p50 %.2f p99 %.2f | nodes %.1f (p90 %.0f max %.0f)  leaves %.1f (p90 %.0f max %.0f)  pts %.0f (p90 %.0f)  stack p50 %d p99 %d max %d [%.1fs]" % (
        name, L, w, T.D, kd_levels, np.sqrt(np.median(res[:, 0])), np.sqrt(np.percentile(res[:, 0], 99)), res[:, 1].mean(), np.percentile(res[:, 1], 90), res[:, 1].max(),
        res[:, 2].mean(), np.percentile(res[:, 2], 90), res[:, 2].max(), res[:, 3].mean(), np.percentile(res[:, 3], 90), np.median(res[:, 4]), np.percentile(res[:, 4], 99), res[:, 4].max(), time.time() - t0), flush=True)
    return T


# ---- the structure as the GPU would build it: top T levels from a strided sample (planes), points routed to 2^T buckets,
# ---- bottom levels balanced inside each bucket; widest dimension from the CELL (parent's box cut at the plane) or the tight box
class GpuTree(BalTree):
    def __init__(self, F, L, w, T, S, cell_based=True):
        n = len(F)
        D = T
        while (n >> D) > L:  # average leaf <= L
            D += 1
        self.D, self.w, self.n, self.T = D, w, n, T
        # --- top tree on the sample
        samp = F[(np.arange(S) * n) // S] if S < n else F.copy()
        planes = {}

        def top(d, i, idx, lo, hi):
            if d == T:
                return
            P = samp[idx]
            if not cell_based and len(P):
                lo, hi = P.min(0), P.max(0)
            dim = int(np.argmax(hi - lo))
            o = np.argsort(P[:, dim], kind="stable")
            idx = idx[o]
            m = len(idx) // 2
            v = samp[idx[m], dim] if len(idx) else 0.5 * (lo[dim] + hi[dim])
            planes[(d, i)] = (dim, v)
            hl, lr = hi.copy(), lo.copy()
            hl[dim], lr[dim] = v, v
            top(d + 1, 2 * i, idx[:m], lo, hl)
            top(d + 1, 2 * i + 1, idx[m:], lr, hi)

        top(0, 0, np.arange(len(samp)), samp.min(0), samp.max(0))
        # --- route
        b = np.zeros(n, np.int64)
        for d in range(T):
            dims = np.array([planes[(d, i)][0] for i in range(1 << d)])
            vals = np.array([planes[(d, i)][1] for i in range(1 << d)])
            right = F[np.arange(n), dims[b]] >= vals[b]
            b = 2 * b + right
        perm = np.argsort(b, kind="stable")
        cnt = np.bincount(b, minlength=1 << T)
        starts = np.concatenate([[0], np.cumsum(cnt)])
        Bd = D - T
        leaf_begin = np.zeros((1 << D) + 1, np.int64)
        self.bucket_sizes = cnt

        def bottom(a0, nb, j, i, lo, hi):
            a, e = a0 + ((nb * i) >> j), a0 + ((nb * (i + 1)) >> j)
            if j == Bd:
                return
            P = F[perm[a:e]]
            if len(P) > 1:
                if not cell_based:
                    lo, hi = P.min(0), P.max(0)
                dim = int(np.argmax(hi - lo))
                o = np.argsort(P[:, dim], kind="stable")
                perm[a:e] = perm[a:e][o]
                m = a0 + ((nb * (2 * i + 1)) >> (j + 1))
                v = F[perm[min(m, e - 1)], dim]
                hl, lr = hi.copy(), lo.copy()
                hl[dim], lr[dim] = v, v
            else:
                hl, lr = hi, lo
            bottom(a0, nb, j + 1, 2 * i, lo, hl)
            bottom(a0, nb, j + 1, 2 * i + 1, lr, hi)

        for bk in range(1 << T):
            a0, nb = starts[bk], cnt[bk]
            if nb:
                P = F[perm[a0:a0 + nb]]
                bottom(a0, nb, 0, 0, P.min(0), P.max(0))
            idx = np.arange((1 << Bd) + 1)
            leaf_begin[(bk << Bd):(bk << Bd) + (1 << Bd) + 1] = a0 + ((nb * idx) >> Bd)
        self.F = F[perm]
        self.perm = perm
        self.leaf_begin = leaf_begin
        big = 1e30
        lo = np.full((1 << D, 6), big)
        hi = np.full((1 << D, 6), -big)
        for i in range(1 << D):
            a, e = leaf_begin[i], leaf_begin[i + 1]
            if e > a:
                lo[i], hi[i] = self.F[a:e].min(0), self.F[a:e].max(0)
        self.lo, self.hi = [None] * (D + 1), [None] * (D + 1)
        self.lo[D], self.hi[D] = lo, hi
        for d in range(D - 1, -1, -1):
            lo = np.minimum(lo[0::2], lo[1::2])
            hi = np.maximum(hi[0::2], hi[1::2])
            self.lo[d], self.hi[d] = lo, hi

    def query(self, q, k, seed_pos=None):
        D, w = self.D, self.w
        best = np.full(k, np.inf)
        st = [0, 0, 0]
        maxstack = 0
        first = D % w if D % w else w
        stack = [(0, 0, 0.0)]
        while stack:
            d, i, dd = stack.pop()
            if dd > best[-1]:
                continue
            dn = d + (first if d == 0 else w)
            c0, c1 = i << (dn - d), (i + 1) << (dn - d)
            lo, hi = self.lo[dn][c0:c1], self.hi[dn][c0:c1]
            dv = np.maximum(np.maximum(lo - q, q - hi), 0.0)
            d2 = (dv * dv).sum(1)
            st[0] += 1
            if dn == D:
                for j in np.argsort(d2, kind="stable"):
                    if d2[j] <= best[-1]:
                        a, e = self.leaf_begin[c0 + j], self.leaf_begin[c0 + j + 1]
                        dist = ((self.F[a:e] - q) ** 2).sum(1)
                        best = np.sort(np.concatenate([best, dist]))[:k]
                        st[1] += 1
                        st[2] += e - a
                continue
            for j in np.argsort(-d2, kind="stable"):
                if d2[j] <= best[-1]:
                    stack.append((dn, c0 + int(j), float(d2[j])))
            maxstack = max(maxstack, len(stack))
        return best[-1], st[0], st[1], st[2], maxstack


def run_gpu(name, Fq, Ft, same, L, w, T, S, cell_based, nsamp=1000, k=10):
    t0 = time.time()
    Tr = GpuTree(Ft, L, w, T, S, cell_based)
    rng = np.random.default_rng(1)
    qs = Ft[rng.choice(len(Ft), nsamp, replace=False)] if same else Fq[rng.choice(len(Fq), nsamp, replace=False)]
    res = np.array([Tr.query(q, k) for q in qs])
    bs = Tr.bucket_sizes
    print("%-12s L=%2d w=%d T=%2d S=%5d D=%2d cell=%d: buckets %d..%d | nodes %.1f (p90 %.0f max %.0f)  leaves %.1f (p90 %.0f max %.0f)  pts %.0f (p90 %.0f)  stack p50 %d p99 %d max %d [%.1fs]" % (
        name, L, w, T, S, Tr.D, cell_based, bs.min(), bs.max(), res[:, 1].mean(), np.percentile(res[:, 1], 90), res[:, 1].max(),
        res[:, 2].mean(), np.percentile(res[:, 2], 90), res[:, 2].max(), res[:, 3].mean(), np.percentile(res[:, 3], 90), np.median(res[:, 4]), np.percentile(res[:, 4], 99), res[:, 4].max(), time.time() - t0), flush=True)
    return Tr
