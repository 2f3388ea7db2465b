// bih_cull_study.cpp -- how many BIH steps of the flagship frame exist only because a BIH node bounds ONE axis?
//
// The reference's tree (Bih.hs:211-285) is built here by the product's own host builder (host_graph.hpp) over the S3 / S5
// heightfield; camera rays of the 1920x1080 frame (sampled every STRIDE-th pixel) and their shadow rays are walked per ray the way
// the kernels walk them (closest hit with ordered early-out, any hit), counting branch steps and triangle tests, under variants:
//   none     the walk as it is
//   slab_b   + every node carries the true extent of its subtree along ONE fixed axis b (the tree's "thin" axis); a ray's interval
//            is clipped with it on entry
//   aabb     + every node carries the true box of its subtree (what a BVH has)
// A cull only ever drops nodes in which nothing can be hit: results are unchanged (checked: same hits).
//
//   g++ -O2 -std=c++17 -I glome_amd/csrc -I include tools/probe/bih_cull_study.cpp -o /tmp/bih_cull_study && /tmp/bih_cull_study 224 8
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "host_graph.hpp"

using namespace glome;

static double sfun(long k) { double u = double(k % 32) / 32.0; double sign = ((k / 32) % 2 == 0) ? 1.0 : -1.0; return sign * (4.0 * u * (1.0 - u)); }

struct Tri { D3 p, e1, e2; };
struct Study {
  const BihTree* T;
  std::vector<Tri> tris;            // by item index
  std::vector<Box3> nbox;           // true box of every node's subtree
  long steps = 0, tests = 0;
  int variant = 0, baxis = 1;
  std::vector<int>* visited = nullptr;  // when set: every branch node entered is appended
  bool tri_hit(const Tri& t, D3 o, D3 d, double tmax, double& tt) const {
    D3 s1 = cross(d, t.e2);
    double div = dot(s1, t.e1);
    if (div == 0) return false;
    double inv = 1.0 / div;
    D3 dd = o - t.p;
    double b1 = dot(dd, s1) * inv;
    if (b1 < 0 || b1 > 1) return false;
    D3 s2 = cross(dd, t.e1);
    double b2 = dot(d, s2) * inv;
    if (b2 < 0 || b1 + b2 > 1) return false;
    tt = dot(t.e2, s2) * inv;
    return !(tt < 0) && !(tt > tmax);
  }
  static double comp(D3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
  // clip [nearv, farv] with node k's bound under the variant; false = empty
  bool clip(int k, D3 o, D3 rcp, double& nearv, double& farv) const {
    if (variant == 0) return true;
    const Box3& b = nbox[k];
    for (int a = 0; a < 3; a++) {
      if (variant == 1 && a != baxis) continue;
      double t0 = (comp(b.lo, a) - comp(o, a)) * comp(rcp, a), t1 = (comp(b.hi, a) - comp(o, a)) * comp(rcp, a);
      if (t0 > t1) std::swap(t0, t1);
      nearv = std::max(nearv, t0); farv = std::min(farv, t1);
    }
    return !(nearv > farv);
  }
  // mode 1: closest (ordered early-out), mode 2: any hit.  returns best t or -1
  double walk(D3 o, D3 d, double dmax, int mode) {
    D3 rcp{1.0 / d.x, 1.0 / d.y, 1.0 / d.z};
    double nearv = 0, farv = dmax;
    {  // root box clip (bbclip_ub)
      const Box3& b = T->bb;
      double tn = -1e300, tf = 1e300;
      for (int a = 0; a < 3; a++) {
        double t0 = (comp(b.lo, a) - comp(o, a)) * comp(rcp, a), t1 = (comp(b.hi, a) - comp(o, a)) * comp(rcp, a);
        if (t0 > t1) std::swap(t0, t1);
        tn = std::max(tn, t0); tf = std::min(tf, t1);
      }
      nearv = tn; farv = std::min(dmax, tf);
      if (nearv > farv) return -1;
    }
    struct E { int k; double n, f; };
    std::vector<E> st;
    int k = 0;
    double best = -1;
    for (;;) {
      bool pop = true;
      if (mode == 1 && best >= 0) farv = std::min(farv, best);
      if (!(nearv > farv) && clip(k, o, rcp, nearv, farv)) {
        const BihTree::Node& n = T->nodes[k];
        if (n.leaf) {
          for (int it : n.items) {
            tests++;
            double tt;
            if (tri_hit(tris[it], o, d, farv, tt)) {
              if (mode == 2) return tt;
              if (best < 0 || !(best < tt)) { best = tt; farv = std::min(farv, tt); }
            }
          }
        } else {
          steps++;
          if (visited) visited->push_back(k);
          double oa = comp(o, n.axis), ra = comp(rcp, n.axis);
          double dl = (n.lsplit - oa) * ra, dr = (n.rsplit - oa) * ra;
          bool fwd = ra > 0;
          int c1 = fwd ? n.left : n.right, c2 = fwd ? n.right : n.left;
          double t1 = fwd ? dl : dr, t2 = fwd ? dr : dl;
          bool e1 = T->nodes[c1].leaf && T->nodes[c1].items.empty(), e2 = T->nodes[c2].leaf && T->nodes[c2].items.empty();
          bool go1 = nearv < t1 && !e1, go2 = t2 < farv && !e2;
          if (go1) {
            if (go2) st.push_back({c2, std::max(t2, nearv), farv});
            k = c1; farv = std::min(t1, farv); pop = false;
          } else if (go2) { k = c2; nearv = std::max(t2, nearv); pop = false; }
        }
      }
      if (pop) {
        if (st.empty()) return best;
        E e = st.back(); st.pop_back();
        k = e.k; nearv = e.n; farv = e.f;
      }
    }
  }
};

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 224, STRIDE = argc > 2 ? atoi(argv[2]) : 8;
  const int W = N > 300 ? 3840 : 1920, H = N > 300 ? 2160 : 1080;
  Graph G;
  std::vector<D3> V((size_t)(N + 1) * (N + 1));
  for (long i = 0; i <= N; i++) for (long j = 0; j <= N; j++) {
    uint32_t h = (uint32_t)((uint64_t)i * 73856093u) ^ (uint32_t)((uint64_t)j * 19349663u);
    double y = 1.5 * sfun(i) * sfun(j + 17) + 0.05 * (double(h & 1023) / 1024.0);
    auto r32 = [](double x) { return (double)(float)x; };
    V[i * (N + 1) + j] = D3{r32(i * 20.0 / N - 10.0), r32(y), r32(j * 20.0 / N - 10.0)};
  }
  std::vector<int> ids;
  std::vector<Tri> tris;
  auto add = [&](D3 a, D3 b, D3 c) { ids.push_back(G.triangle(a, b, c)); tris.push_back({a, b - a, c - a}); };
  for (long i = 0; i < N; i++) for (long j = 0; j < N; j++) {
    D3 a = V[i * (N + 1) + j], b = V[i * (N + 1) + j + 1], c = V[(i + 1) * (N + 1) + j], d = V[(i + 1) * (N + 1) + j + 1];
    add(a, b, c); add(c, b, d);
  }
  int root = G.bih(ids);
  const BihTree& T = *G.at(root).bih;
  // the items of a leaf are indices into `ids` order?  BihTree::Node::items holds graph node ids: map them back
  std::vector<int> of_id(G.nodes.size(), -1);
  for (size_t k = 0; k < ids.size(); k++) of_id[ids[k]] = (int)k;
  BihTree T2 = T;
  for (auto& n : T2.nodes) for (auto& it : n.items) it = of_id[it];
  Study S; S.T = &T2; S.tris = tris;
  S.nbox.assign(T2.nodes.size(), box_empty());
  long nleaf = 0, nbranch = 0, axis_count[3] = {0, 0, 0};
  for (int k = (int)T2.nodes.size() - 1; k >= 0; k--) {  // preorder array: children come after their parent
    const auto& n = T2.nodes[k];
    if (n.leaf) { nleaf++; for (int it : n.items) { const Tri& t = tris[it]; D3 pts[3] = {t.p, t.p + t.e1, t.p + t.e2}; S.nbox[k] = box_join(S.nbox[k], box_of_points(pts, 3)); } }
    else { nbranch++; axis_count[n.axis]++; S.nbox[k] = box_join(S.nbox[n.left], S.nbox[n.right]); }
  }
  printf("tree: %zu nodes (%ld branches: x %ld y %ld z %ld; %ld leaves), depth %d\n", T2.nodes.size(), nbranch, axis_count[0], axis_count[1], axis_count[2], nleaf, T2.depth);
  // camera (Scene.hs:48-57) and the one light of S3
  D3 pos{-2, 4.3, 15}, at{0, 2, 0}, up{0, 1, 0}, light{-100, 70, 140};
  D3 fwd = normalize(at - pos), right = normalize(cross(up, fwd)), up_ = normalize(cross(fwd, right));
  double cs = std::tan((M_PI / 180) * 22.5);
  up_ = up_ * cs; right = right * cs;
  const char* names[3] = {"none", "slab_y", "aabb"};
  std::vector<double> ref_t;
  for (int v = 0; v < 3; v++) {
    S.variant = v; S.baxis = 1;
    long psteps = 0, ptests = 0, ssteps = 0, stests = 0, nprim = 0, nshadow = 0, nocc = 0;
    size_t idx = 0;
    for (int py = STRIDE / 2; py < H; py += STRIDE) for (int px = STRIDE / 2; px < W; px += STRIDE) {
      double xc = ((double(px) / W) * 2 - 1) * (double(W) / H), yc = -((double(py) / H) * 2 - 1);
      D3 d = normalize(fwd + right * (-xc) + up_ * yc);
      S.steps = S.tests = 0;
      double t = S.walk(pos, d, 1e6, 1);
      psteps += S.steps; ptests += S.tests; nprim++;
      if (v == 0) ref_t.push_back(t); else if (ref_t[idx] != t) { printf("MISMATCH at %d %d: %g vs %g\n", px, py, ref_t[idx], t); }
      idx++;
      if (t >= 0) {
        D3 p = pos + d * t;
        const Tri* hit = nullptr; (void)hit;
        // the normal: +y-ish for a heightfield; the kernels use the triangle's; the study only needs the shadow ray's origin off the surface
        D3 n{0, 1, 0};
        D3 lv = light - p;
        double ll = std::sqrt(dot(lv, lv));
        D3 ld = lv * (1.0 / ll);
        S.steps = S.tests = 0;
        double ts = S.walk(p + n * 1e-4, ld, ll - 2e-4, 2);
        ssteps += S.steps; stests += S.tests; nshadow++; nocc += ts >= 0;
      }
    }
    printf("%-7s primary: %ld rays, %.1f steps %.1f tests per ray | shadow: %ld rays (%ld occluded), %.1f steps %.1f tests per ray | all: %.1f steps %.1f tests per ray\n", names[v], nprim,
           double(psteps) / nprim, double(ptests) / nprim, nshadow, nocc, double(ssteps) / std::max(1L, nshadow), double(stests) / std::max(1L, nshadow),
           double(psteps + ssteps) / (nprim + nshadow), double(ptests + stests) / (nprim + nshadow));
  }
  // ---- per PACKET (one 8x8 pixel block = one work item): branch nodes entered by any of its 64 rays (what the wave-wide walk steps
  // through), primary walk and shadow walk, over every BSTRIDE-th block of the frame
  {
    S.variant = 0;
    const int BSTRIDE = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<long> pu, su, ptot;
    std::vector<int> vis, all;
    for (int by = 0; by + 8 <= H; by += 8 * BSTRIDE) for (int bx = 0; bx + 8 <= W; bx += 8 * BSTRIDE) {
      std::vector<D3> so, sdv; std::vector<double> sl;
      all.clear();
      long own_max = 0;
      for (int j = 0; j < 64; j++) {
        int px = bx + j % 8, py = by + j / 8;
        double xc = ((double(px) / W) * 2 - 1) * (double(W) / H), yc = -((double(py) / H) * 2 - 1);
        D3 d = normalize(fwd + right * (-xc) + up_ * yc);
        vis.clear(); S.visited = &vis;
        double t = S.walk(pos, d, 1e6, 1);
        all.insert(all.end(), vis.begin(), vis.end());
        if (t >= 0) { D3 p = pos + d * t; D3 lv = light - p; double ll = std::sqrt(dot(lv, lv)); so.push_back(p + D3{0, 1e-4, 0}); sdv.push_back(lv * (1.0 / ll)); sl.push_back(ll - 2e-4); }
      }
      std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
      long pun = (long)all.size();
      all.clear();
      for (size_t j = 0; j < so.size(); j++) {
        vis.clear(); S.visited = &vis;
        S.walk(so[j], sdv[j], sl[j], 2);
        own_max = std::max<long>(own_max, (long)vis.size());
        all.insert(all.end(), vis.begin(), vis.end());
      }
      std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
      pu.push_back(pun); su.push_back((long)all.size()); ptot.push_back(pun + (long)all.size());
      (void)own_max;
    }
    S.visited = nullptr;
    auto q = [](std::vector<long> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    auto mean = [](const std::vector<long>& v) { double s = 0; for (long x : v) s += x; return s / v.size(); };
    printf("packets (%zu sampled): branch steps per item  primary walk mean %.0f p50 %ld p99 %ld max %ld | shadow walk mean %.0f p50 %ld p90 %ld p99 %ld p999 %ld max %ld | item mean %.0f p50 %ld p99 %ld max %ld (max / mean %.1f)\n",
           pu.size(), mean(pu), q(pu, .5), q(pu, .99), q(pu, 1.0), mean(su), q(su, .5), q(su, .9), q(su, .99), q(su, .999), q(su, 1.0), mean(ptot), q(ptot, .5), q(ptot, .99), q(ptot, 1.0), q(ptot, 1.0) / mean(ptot));
  }
  return 0;
}
