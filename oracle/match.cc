// oracle/match.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restatement of the reference surfel matcher:
//   ToVector            src/odometry/knn_surfel_matcher.cc:91-98   (6-D feature: center/1.0, normal/5deg, world frame)
//   BuildIndex          src/odometry/knn_surfel_matcher.cc:3-14
//   FLANNKNearestSearch src/odometry/knn_surfel_matcher.cc:75-89   (exact: checks = -1, eps = 0, squared L2, sorted)
//   Match               src/odometry/knn_surfel_matcher.cc:16-49
//   AngularDistance     src/odometry/surfel.h:105-107
// FLANN 1.9.1 (un-vendored, pulled in through PCL) is replaced by an own exact kd-tree (leaf size 15 like
// KDTreeSingleIndexParams(15)); any exact search returns the same neighbours up to distance ties, which are
// broken here by ascending index.  Pinned by the reference's own property test
// (knn_surfel_matcher_test.cc:19-43) in tests/test_oracle_kat.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <set>
#include <utility>
#include <vector>

#include "math3.h"
#include "wc_oracle.h"

namespace {
using namespace wco;

struct KdTree {
  static constexpr int D = 6;
  static constexpr int LEAF = 15;
  struct Node {
    int lo, hi;       // point range in perm
    int left, right;  // children (-1 for leaf)
    int dim;
    double split_lo, split_hi;  // max of left side / min of right side along dim
  };
  const double *pts;
  int n;
  std::vector<int> perm;
  std::vector<Node> nodes;

  void build(const double *p, int count) {
    pts = p;
    n = count;
    perm.resize(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    nodes.clear();
    if (n > 0) build_rec(0, n);
  }
  int build_rec(int lo, int hi) {
    int id = (int)nodes.size();
    nodes.push_back({lo, hi, -1, -1, 0, 0, 0});
    if (hi - lo <= LEAF) return id;
    double mn[D], mx[D];
    for (int d = 0; d < D; ++d) mn[d] = 1e300, mx[d] = -1e300;
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < D; ++d) {
        double v = pts[(size_t)perm[i] * D + d];
        mn[d] = std::min(mn[d], v);
        mx[d] = std::max(mx[d], v);
      }
    int dim = 0;
    for (int d = 1; d < D; ++d)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (mx[dim] == mn[dim]) return id;  // all identical: keep as (big) leaf
    int mid = (lo + hi) / 2;
    std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, [&](int a, int b) {
      double va = pts[(size_t)a * D + dim], vb = pts[(size_t)b * D + dim];
      return va < vb || (va == vb && a < b);
    });
    double slo = -1e300, shi = 1e300;
    for (int i = lo; i < mid; ++i) slo = std::max(slo, pts[(size_t)perm[i] * D + dim]);
    for (int i = mid; i < hi; ++i) shi = std::min(shi, pts[(size_t)perm[i] * D + dim]);
    int l = build_rec(lo, mid);
    int r = build_rec(mid, hi);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].dim = dim;
    nodes[id].split_lo = slo;
    nodes[id].split_hi = shi;
    return id;
  }

  struct Best {
    int k, cnt;
    int32_t *idx;
    double *d2;
    double worst() const { return cnt < k ? 1e300 : d2[k - 1]; }
    void push(int i, double d) {
      if (cnt == k && !(d < d2[k - 1] || (d == d2[k - 1] && i < idx[k - 1]))) return;
      int pos = cnt < k ? cnt++ : k - 1;
      while (pos > 0 && (d2[pos - 1] > d || (d2[pos - 1] == d && idx[pos - 1] > i))) {
        d2[pos] = d2[pos - 1];
        idx[pos] = idx[pos - 1];
        --pos;
      }
      d2[pos] = d;
      idx[pos] = i;
    }
  };

  void search(const double *q, Best &b, int node) const {
    const Node &nd = nodes[node];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        const double *p = pts + (size_t)perm[i] * D;
        double s = 0;
        for (int d = 0; d < D; ++d) {  // L2_Simple: plain sum of squared differences
          double df = q[d] - p[d];
          s += df * df;
        }
        b.push(perm[i], s);
      }
      return;
    }
    double v = q[nd.dim];
    // distance from q to each child's slab along the split dimension
    double dl = v > nd.split_lo ? v - nd.split_lo : 0.0;
    double dr = v < nd.split_hi ? nd.split_hi - v : 0.0;
    int first = dl <= dr ? nd.left : nd.right, second = dl <= dr ? nd.right : nd.left;
    double dfirst = std::min(dl, dr), dsecond = std::max(dl, dr);
    if (dfirst * dfirst <= b.worst()) search(q, b, first);
    if (dsecond * dsecond <= b.worst()) search(q, b, second);
  }
};

void feature6(const wc_params *P, const wc_surfel &s, const wc_pose &ps, double out[6], V3 *cw, V3 *nw) {
  Q4 rot{ps.quat[0], ps.quat[1], ps.quat[2], ps.quat[3]};
  V3 pos{ps.pos[0], ps.pos[1], ps.pos[2]};
  V3 c = qrot(rot, V3{s.center[0], s.center[1], s.center[2]}) + pos;  // GetCenterInWorld, surfel.h:67-69
  V3 nn = qrot(rot, V3{s.normal[0], s.normal[1], s.normal[2]});       // GetNormInWorld,   surfel.h:78-80
  V3 cu = c / P->center_scale, nu = nn / P->angular_scale;
  out[0] = cu.x, out[1] = cu.y, out[2] = cu.z, out[3] = nu.x, out[4] = nu.y, out[5] = nu.z;
  if (cw) *cw = c;
  if (nw) *nw = nn;
}
}  // namespace

extern "C" int wco_knn6(const double *cloud6, uint64_t n, const double *query6, uint64_t nq, int k, int32_t *idx,
                        double *dist2) {
  KdTree tree;
  tree.build(cloud6, (int)n);
  for (uint64_t q = 0; q < nq; ++q) {
    KdTree::Best b{k, 0, idx + q * k, dist2 + q * k};
    if (n > 0) tree.search(query6 + q * 6, b, 0);
    for (int i = b.cnt; i < k; ++i) {  // Q10: FLANN leaves the tail untouched (zero-initialised vector)
      b.idx[i] = 0;
      b.d2[i] = 0;
    }
  }
  return 0;
}

extern "C" int wco_match(const wc_params *P, const wc_surfel *q_surf, const wc_pose *q_pose, uint64_t nq,
                         const wc_surfel *t_surf, const wc_pose *t_pose, uint64_t nt, int same_set, wc_pair *pairs,
                         uint64_t cap, uint64_t *n_pairs) {
  *n_pairs = 0;
  if (nt == 0) return 0;  // knn_surfel_matcher.cc:18-20
  const int k = P->knn_k;
  std::vector<double> cloud(nt * 6);
  std::vector<V3> tc(nt), tn(nt);
  for (uint64_t i = 0; i < nt; ++i) feature6(P, t_surf[i], t_pose[i], &cloud[i * 6], &tc[i], &tn[i]);
  KdTree tree;
  tree.build(cloud.data(), (int)nt);

  std::set<std::pair<int64_t, int64_t>> seen;  // std::set of (query, candidate) identities, cc:21
  std::vector<int32_t> idx(k);
  std::vector<double> d2(k);
  uint64_t out = 0;
  for (uint64_t q = 0; q < nq; ++q) {
    double f[6];
    V3 cq, nq_w;
    feature6(P, q_surf[q], q_pose[q], f, &cq, &nq_w);
    KdTree::Best b{k, 0, idx.data(), d2.data()};
    tree.search(f, b, 0);
    for (int i = b.cnt; i < k; ++i) idx[i] = 0;  // Q10
    // identities: in same_set mode query q and target q are the same object
    const int64_t qid = same_set ? (int64_t)q : -(int64_t)q - 1;
    for (int j = 0; j < k; ++j) {
      const int c = idx[j];
      if (std::fabs(t_surf[c].t - q_surf[q].t) < P->time_diff_min) continue;          // cc:26
      if (std::acos(dot(nq_w, tn[c])) > P->angular_scale) continue;                      // cc:29
      if (std::fabs(dot(nq_w, cq - tc[c])) > P->surfel_dist_max) continue;               // cc:32
      if (seen.count({qid, (int64_t)c}) || seen.count({(int64_t)c, qid})) continue;      // cc:35-38
      seen.insert({qid, (int64_t)c});
      if (out < cap) {
        if (same_set) {
          if (q_surf[q].t < t_surf[c].t)
            pairs[out] = {(int32_t)q, c};
          else
            pairs[out] = {c, (int32_t)q};
        } else {
          // fixed-window matcher: the target must be the older one (CHECK_LT at lidar_odometry.cc:301)
          if (!(t_surf[c].t < q_surf[q].t)) return 3;
          pairs[out] = {c, (int32_t)q};
        }
      }
      ++out;
      break;
    }
  }
  *n_pairs = out;
  return out > cap ? 1 : 0;
}
