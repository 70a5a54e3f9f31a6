// oracle/extract.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Single-thread CPU restatement of the reference surfel extraction, with the reference's data-structure
// choices (hash map of root voxels -> per-node point vectors -> 3-level octree -> temporal clusters):
//   BuildSurfels        src/odometry/surfel_extraction.cc:316-337
//   BuildVoxelMap       src/odometry/surfel_extraction.cc:186-220
//   VoxelLoc            src/odometry/surfel_extraction.h:55-64
//   InitOctoTree        src/odometry/surfel_extraction.cc:128-140
//   CutOctoTree         src/odometry/surfel_extraction.cc:142-184
//   InitPlane           src/odometry/surfel_extraction.cc:82-126
//   ExtractSurfelInfo   src/odometry/surfel_extraction.cc:304-314
//   ClusterSurfels      src/odometry/surfel_extraction.cc:12-65
// Decision rules: SURVEY.md Appendix A.  Parity status: UNPINNED against the real reference for this
// stage (the reference has no test or fixture for it and cannot be built here — no Eigen/PCL/absl);
// cross-checked instead by an independent numpy restatement (oracle/np_check.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "math3.h"
#include "wc_oracle.h"

namespace {
using namespace wco;

struct Pt {
  double t;
  V3 p;
};

struct Key {
  int32_t x, y, z;
  bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct KeyHash {
  size_t operator()(const Key &k) const {
    // own 64-bit mix (the reference hash only decides bucket placement, never results)
    uint64_t h = (uint64_t)(uint32_t)k.x * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)k.y + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)(uint32_t)k.z + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
    return (size_t)(h ^ (h >> 29));
  }
};

struct Moments {
  double n = 0, st = 0;
  V3 sp{0, 0, 0};
  M3 spp = M3::zero();
};

struct Pca {
  V3 center;
  M3 cov;
  double ev[3];
  M3 evec;
  double t_mean;
  double likeness;
};

// moments -> mean / un-centred population covariance / eigen-decomposition
// (surfel_extraction.cc:36-51 and :89-101 share this arithmetic)
Pca pca_of(const std::vector<Pt> &pts) {
  M3 acc = M3::zero();
  V3 c{0, 0, 0};
  double ts = 0;
  for (const Pt &q : pts) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) acc.m[i][j] += q.p[i] * q.p[j];
    c = c + q.p;
    ts += q.t;
  }
  const int n = (int)pts.size();
  Pca r;
  r.center = c / (double)n;
  r.t_mean = ts / (double)n;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.cov.m[i][j] = acc.m[i][j] / (double)n - r.center[i] * r.center[j];
  eig3_sym(r.cov, r.ev, r.evec);
  r.likeness = 2 * (r.ev[1] - r.ev[0]) / ((r.ev[0] + r.ev[1]) + r.ev[2]);
  return r;
}

struct Node {
  std::vector<Pt> pts;
  Node *child[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double center[3];
  float quarter;
  int layer;
  bool is_plane = false;
  bool tested = false;
  double margin = 1e300;  // distance of the gate quantities from their thresholds
  ~Node() {
    for (Node *c : child) delete c;
  }
};

struct Ctx {
  const wc_params *P;
  double thr;  // (double)planer_threshold
  wco_extract_stats st;
};

// InitPlane gate (surfel_extraction.cc:106-111)
void test_node(Ctx &cx, Node *nd) {
  Pca r = pca_of(nd->pts);
  nd->tested = true;
  nd->is_plane = (r.ev[0] < cx.thr) && (r.likeness > cx.P->min_plane_likeness);
  nd->margin = std::min(std::fabs(r.ev[0] - cx.thr), std::fabs(r.likeness - cx.P->min_plane_likeness));
  cx.st.nodes_tested[nd->layer]++;
  if (nd->is_plane) cx.st.nodes_plane[nd->layer]++;
  if (nd->margin < cx.st.min_gate_margin) cx.st.min_gate_margin = nd->margin;
}

// CutOctoTree (surfel_extraction.cc:142-184)
void split_node(Ctx &cx, Node *nd) {
  if (nd->layer >= cx.P->max_layer) return;
  for (const Pt &q : nd->pts) {
    int b[3] = {q.p.x > nd->center[0] ? 1 : 0, q.p.y > nd->center[1] ? 1 : 0, q.p.z > nd->center[2] ? 1 : 0};
    int oct = 4 * b[0] + 2 * b[1] + b[2];
    Node *&c = nd->child[oct];
    if (!c) {
      c = new Node;
      c->layer = nd->layer + 1;
      for (int a = 0; a < 3; ++a) {
        float off = (float)(2 * b[a] - 1) * nd->quarter;  // int * float -> float
        c->center[a] = nd->center[a] + (double)off;       // double + float -> double
      }
      c->quarter = nd->quarter / 2;
    }
    c->pts.push_back(q);
  }
  for (Node *c : nd->child) {
    if (!c) continue;
    if ((int)c->pts.size() > cx.P->min_points) {
      test_node(cx, c);
      if (!c->is_plane) split_node(cx, c);
    }
  }
}

// ClusterSurfels (surfel_extraction.cc:12-65)
void cluster_node(Ctx &cx, const Node *nd, Key key, uint32_t node_code, std::vector<wc_surfel> &out,
                  std::vector<wc_surfel_id> &ids) {
  std::vector<std::vector<Pt>> clusters;
  clusters.push_back({nd->pts[0]});
  for (size_t i = 1; i < nd->pts.size(); ++i) {
    if (nd->pts[i].t - clusters.back().back().t > cx.P->cluster_gap)
      clusters.push_back({nd->pts[i]});
    else
      clusters.back().push_back(nd->pts[i]);
  }
  const double resolution = (double)(nd->quarter * 4);  // float * int -> float -> double
  uint32_t ci = 0;
  for (const auto &cl : clusters) {
    uint32_t this_ci = ci++;
    cx.st.clusters_total++;
    if ((int)cl.size() < cx.P->cluster_min_points) continue;
    Pca r = pca_of(cl);
    double m = std::min(std::fabs(r.ev[0] - cx.thr), std::fabs(r.likeness - cx.P->min_plane_likeness));
    if (m < cx.st.min_gate_margin) cx.st.min_gate_margin = m;
    if (r.ev[0] > cx.thr || r.likeness < cx.P->min_plane_likeness) {
      cx.st.clusters_rejected++;
      continue;
    }
    V3 nrm{r.evec.m[0][0], r.evec.m[1][0], r.evec.m[2][0]};
    V3 view{cx.P->view_point[0], cx.P->view_point[1], cx.P->view_point[2]};
    if (dot(nrm, r.center - view) < 0) nrm = -nrm;
    wc_surfel s;
    s.t = r.t_mean;
    for (int i = 0; i < 3; ++i) {
      s.center[i] = r.center[i];
      s.normal[i] = nrm[i];
      for (int j = 0; j < 3; ++j) s.cov[3 * i + j] = r.cov.m[i][j];
    }
    s.resolution = resolution;
    s.sigma = std::sqrt(r.ev[0]);
    out.push_back(s);
    ids.push_back({key.x, key.y, key.z, node_code | (this_ci << 8)});
  }
}

// ExtractSurfelInfo (surfel_extraction.cc:304-314)
void emit_node(Ctx &cx, const Node *nd, Key key, uint32_t code, std::vector<wc_surfel> &out,
               std::vector<wc_surfel_id> &ids) {
  if (nd->is_plane) cluster_node(cx, nd, key, code, out, ids);
  for (int o = 0; o < 8; ++o) {
    const Node *c = nd->child[o];
    if (!c) continue;
    uint32_t cc = (uint32_t)c->layer;
    if (c->layer == 1)
      cc |= (uint32_t)o << 2;
    else
      cc |= (code & (7u << 2)) | ((uint32_t)o << 5);
    emit_node(cx, c, key, cc, out, ids);
  }
}

inline void load_point(const wc_points *pts, uint64_t i, Pt &q) {
  const float *f = (const float *)((const char *)pts->xyz + i * pts->xyz_stride);
  q.p = {(double)f[0], (double)f[1], (double)f[2]};
  std::memcpy(&q.t, (const char *)pts->time + i * pts->time_stride, sizeof(double));
}

inline Key key_of(V3 p, double vs) {
  return {(int32_t)std::floor(p.x / vs), (int32_t)std::floor(p.y / vs), (int32_t)std::floor(p.z / vs)};
}
}  // namespace

extern "C" int wco_voxel_keys(const wc_points *pts, const wc_params *P, int32_t *keys_xyz) {
  const double vs = (double)P->voxel_size;
  for (uint64_t i = 0; i < pts->n; ++i) {
    Pt q;
    load_point(pts, i, q);
    Key k = key_of(q.p, vs);
    keys_xyz[3 * i + 0] = k.x;
    keys_xyz[3 * i + 1] = k.y;
    keys_xyz[3 * i + 2] = k.z;
  }
  return 0;
}

extern "C" int wco_extract_surfels(const wc_points *pts, const wc_params *P, wc_surfel *out, wc_surfel_id *out_ids,
                                   uint64_t cap, uint64_t *n_out, wco_extract_stats *stats) {
  Ctx cx;
  cx.P = P;
  cx.thr = (double)P->planer_threshold;
  std::memset(&cx.st, 0, sizeof(cx.st));
  cx.st.min_gate_margin = 1e300;
  const double vs = (double)P->voxel_size;

  // BuildVoxelMap (surfel_extraction.cc:186-215): bin into root voxels, creation order kept only so that the
  // traversal below is deterministic; results never depend on it (output is re-sorted).
  std::unordered_map<Key, Node *, KeyHash> map;
  std::vector<std::pair<Key, Node *>> roots;
  for (uint64_t i = 0; i < pts->n; ++i) {
    Pt q;
    load_point(pts, i, q);
    Key k = key_of(q.p, vs);
    auto it = map.find(k);
    Node *nd;
    if (it == map.end()) {
      nd = new Node;
      nd->layer = 0;
      nd->quarter = P->voxel_size / 4;
      nd->center[0] = (0.5 + k.x) * P->voxel_size;
      nd->center[1] = (0.5 + k.y) * P->voxel_size;
      nd->center[2] = (0.5 + k.z) * P->voxel_size;
      map.emplace(k, nd);
      roots.push_back({k, nd});
    } else {
      nd = it->second;
    }
    nd->pts.push_back(q);
  }
  cx.st.root_voxels = roots.size();

  // InitOctoTree (surfel_extraction.cc:128-140): test when n > threshold, split regardless (Q4)
  for (auto &kr : roots) {
    Node *nd = kr.second;
    if ((int)nd->pts.size() > P->min_points) {
      test_node(cx, nd);
      split_node(cx, nd);
    }
  }

  std::vector<wc_surfel> sf;
  std::vector<wc_surfel_id> ids;
  for (auto &kr : roots) emit_node(cx, kr.second, kr.first, 0u, sf, ids);
  for (auto &kr : roots) delete kr.second;

  // std::sort by timestamp (surfel_extraction.cc:334); ties resolved canonically by id
  std::vector<uint32_t> order(sf.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (sf[a].t != sf[b].t) return sf[a].t < sf[b].t;
    const wc_surfel_id &x = ids[a], &y = ids[b];
    if (x.kx != y.kx) return x.kx < y.kx;
    if (x.ky != y.ky) return x.ky < y.ky;
    if (x.kz != y.kz) return x.kz < y.kz;
    return x.node < y.node;
  });
  cx.st.surfels = sf.size();
  if (stats) *stats = cx.st;
  *n_out = sf.size();
  if (sf.size() > cap) return 1;
  for (size_t i = 0; i < order.size(); ++i) {
    out[i] = sf[order[i]];
    if (out_ids) out_ids[i] = ids[order[i]];
  }
  return 0;
}
