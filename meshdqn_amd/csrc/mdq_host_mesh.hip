// Host-side (C++) mesh engine of the batched environment step: vertex removal + re-triangulation,
// DOLFIN-style smoothing and Taylor-Hood topology for B environments at once (std::thread over envs).
//
// Reference semantics (Env2DAirfoil.py:452-512 `_remove_vertex`, flow_solver.py:233-250 `remesh`):
//   drop the vertex, Delaunay-triangulate ALL remaining points (scipy/Qhull), drop the simplices made of
//   boundary vertices only, rebuild the mesh, smooth(50), recompute boundary / removable flags.
// Here the SAME triangulation is obtained without a global rebuild: the Delaunay triangulation of a point
// set in general position is unique, so re-triangulating the removed vertex' star (ear clipping) and
// applying Lawson flips to every interior edge until all are locally Delaunay yields it (the fluid
// boundary edges are Delaunay edges of the full point set on these meshes - SURVEY.md 8(a7) - so the
// constrained and the filtered unconstrained triangulations coincide).  Written with flat arrays and
// no allocation in the inner loops: this is the algorithm the GPU port of the next round follows.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

#include <condition_variable>
#include <functional>
#include <mutex>

namespace mdq_host {

// Persistent worker pool (created on first use, sized on demand): the batched engine is called once per
// environment step, so per-call std::thread creation (~30 us each) would dominate at B = 128.
class Pool {
 public:
  static Pool& get() {
    static Pool p;
    return p;
  }
  // run fn(i) for i in [0, n) on up to `nthreads` workers (the caller participates)
  void parallel_for(int n, int nthreads, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (nthreads <= 1 || n == 1) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    std::unique_lock<std::mutex> call_lock(call_mu_);  // one parallel_for at a time
    ensure(nthreads - 1);
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn;
      next_ = 0;
      n_ = n;
      active_ = std::min<int>(nthreads - 1, (int)workers_.size());
      pending_ = active_;
      ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  Pool() = default;
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
      ++epoch_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void ensure(int k) {
    while ((int)workers_.size() < k) {
      const int id = (int)workers_.size();
      workers_.emplace_back([this, id] {
        uint64_t seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && id < active_); });
            if (stop_) return;
            seen = epoch_;
          }
          work();
          {
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_all();
          }
        }
      });
    }
  }
  void work() {
    for (;;) {
      int i;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (next_ >= n_) return;
        i = next_++;
      }
      (*fn_)(i);
    }
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, n_ = 0, active_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

struct EdgeMap {  // open addressing hash: key (a<b) -> value
  std::vector<int64_t> key;
  std::vector<int32_t> val;
  uint32_t mask;
  explicit EdgeMap(int n) {
    uint32_t cap = 16;
    while (cap < (uint32_t)(2 * n + 8)) cap <<= 1;
    key.assign(cap, -1);
    val.assign(cap, -1);
    mask = cap - 1;
  }
  static int64_t mk(int a, int b) { return a < b ? ((int64_t)a << 32) | (uint32_t)b : ((int64_t)b << 32) | (uint32_t)a; }
  int32_t* find_or_insert(int a, int b, bool& inserted) {
    const int64_t k = mk(a, b);
    uint32_t h = (uint32_t)((uint64_t)k * 0x9E3779B97F4A7C15ull >> 40) & mask;
    while (key[h] != -1 && key[h] != k) h = (h + 1) & mask;
    inserted = key[h] == -1;
    key[h] = k;
    return &val[h];
  }
  int32_t* find(int a, int b) {
    const int64_t k = mk(a, b);
    uint32_t h = (uint32_t)((uint64_t)k * 0x9E3779B97F4A7C15ull >> 40) & mask;
    while (key[h] != -1 && key[h] != k) h = (h + 1) & mask;
    return key[h] == -1 ? nullptr : &val[h];
  }
};

static inline double orient2d(const double* a, const double* b, const double* c) {
  return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

// > 0 when d lies inside the circumcircle of the CCW triangle (a,b,c)
static inline double incircle(const double* a, const double* b, const double* c, const double* d) {
  const double ax = a[0] - d[0], ay = a[1] - d[1], bx = b[0] - d[0], by = b[1] - d[1], cx = c[0] - d[0],
               cy = c[1] - d[1];
  return (ax * ax + ay * ay) * (bx * cy - by * cx) - (bx * bx + by * by) * (ax * cy - ay * cx) +
         (cx * cx + cy * cy) * (ax * by - ay * bx);
}

struct Mesh {
  int nv, nt;
  double* x;     // [nv][2]
  int32_t* tri;  // [nt][3]
};

// make every triangle counter-clockwise
static void orient_ccw(Mesh& m) {
  for (int t = 0; t < m.nt; ++t) {
    int32_t* v = m.tri + 3 * t;
    if (orient2d(m.x + 2 * v[0], m.x + 2 * v[1], m.x + 2 * v[2]) < 0) std::swap(v[1], v[2]);
  }
}

// Remove interior vertex `rv`: ear-clip its star polygon.  Returns 0 ok, <0 failure (mesh untouched).
static int remove_vertex(Mesh& m, int rv) {
  int star[64], ns = 0;
  for (int t = 0; t < m.nt; ++t) {
    const int32_t* v = m.tri + 3 * t;
    if (v[0] == rv || v[1] == rv || v[2] == rv) {
      if (ns == 64) return -1;
      star[ns++] = t;
    }
  }
  if (ns < 3) return -2;
  // ring edges (a -> b) with rv on the left (triangles are CCW: (rv, a, b) in cyclic order)
  int ea[64], eb[64];
  for (int s = 0; s < ns; ++s) {
    const int32_t* v = m.tri + 3 * star[s];
    const int k = v[0] == rv ? 0 : (v[1] == rv ? 1 : 2);
    ea[s] = v[(k + 1) % 3];
    eb[s] = v[(k + 2) % 3];
  }
  int ring[64];
  ring[0] = ea[0];
  int cur = eb[0];
  for (int n = 1; n < ns; ++n) {
    ring[n] = cur;
    int f = -1;
    for (int s = 0; s < ns; ++s)
      if (ea[s] == cur) f = s;
    if (f < 0) return -3;  // open star: rv is a boundary vertex
    cur = eb[f];
  }
  if (cur != ring[0]) return -3;
  // ear clipping of the CCW polygon ring[0..ns)
  int poly[64], np_ = ns, newtri[62][3], nn = 0;
  std::memcpy(poly, ring, sizeof(int) * ns);
  int guard = 0;
  while (np_ > 3 && guard++ < 4096) {
    bool clipped = false;
    for (int i = 0; i < np_ && !clipped; ++i) {
      const int p0 = poly[(i + np_ - 1) % np_], p1 = poly[i], p2 = poly[(i + 1) % np_];
      const double* A = m.x + 2 * p0;
      const double* Bp = m.x + 2 * p1;
      const double* Cp = m.x + 2 * p2;
      if (orient2d(A, Bp, Cp) <= 0) continue;  // reflex corner
      bool empty = true;
      for (int j = 0; j < np_ && empty; ++j) {
        const int q = poly[j];
        if (q == p0 || q == p1 || q == p2) continue;
        const double* Q = m.x + 2 * q;
        if (orient2d(A, Bp, Q) >= 0 && orient2d(Bp, Cp, Q) >= 0 && orient2d(Cp, A, Q) >= 0) empty = false;
      }
      if (!empty) continue;
      newtri[nn][0] = p0;
      newtri[nn][1] = p1;
      newtri[nn][2] = p2;
      ++nn;
      for (int j = i; j + 1 < np_; ++j) poly[j] = poly[j + 1];
      --np_;
      clipped = true;
    }
    if (!clipped) return -4;
  }
  if (np_ != 3) return -4;
  newtri[nn][0] = poly[0];
  newtri[nn][1] = poly[1];
  newtri[nn][2] = poly[2];
  ++nn;  // nn == ns - 2
  // write the new triangles into the first nn star slots, compact the last two away
  for (int s = 0; s < nn; ++s) std::memcpy(m.tri + 3 * star[s], newtri[s], sizeof(int32_t) * 3);
  int dead[2] = {star[ns - 2], star[ns - 1]};
  if (dead[0] < dead[1]) std::swap(dead[0], dead[1]);  // remove the higher slot first
  for (int q = 0; q < 2; ++q) {
    const int last = m.nt - 1;
    if (dead[q] != last) std::memcpy(m.tri + 3 * dead[q], m.tri + 3 * last, sizeof(int32_t) * 3);
    --m.nt;
  }
  // drop the vertex: ids above shift down
  for (int i = 0; i < 3 * m.nt; ++i)
    if (m.tri[i] > rv) --m.tri[i];
  std::memmove(m.x + 2 * rv, m.x + 2 * (rv + 1), sizeof(double) * 2 * (m.nv - rv - 1));
  --m.nv;
  return 0;
}

// Lawson flips until every interior edge is locally Delaunay.  Returns number of flips, <0 on failure.
static int make_delaunay(Mesh& m) {
  const int nt = m.nt;
  std::vector<int32_t> nbr(3 * nt, -1);  // nbr[3t+k]: triangle across the edge opposite local vertex k
  {
    EdgeMap em(3 * nt);
    for (int t = 0; t < nt; ++t)
      for (int k = 0; k < 3; ++k) {
        const int a = m.tri[3 * t + (k + 1) % 3], b = m.tri[3 * t + (k + 2) % 3];
        bool ins;
        int32_t* slot = em.find_or_insert(a, b, ins);
        if (ins) {
          *slot = 3 * t + k;
        } else {
          const int o = *slot;
          if (nbr[o] != -1) return -1;  // non-manifold
          nbr[o] = t;
          nbr[3 * t + k] = o / 3;
        }
      }
  }
  std::vector<int32_t> stack;
  stack.reserve(3 * nt);
  for (int t = 0; t < nt; ++t)
    for (int k = 0; k < 3; ++k)
      if (nbr[3 * t + k] > t) stack.push_back(3 * t + k);
  int flips = 0;
  long guard = 0;
  while (!stack.empty()) {
    if (++guard > 200000) return -2;
    const int he = stack.back();
    stack.pop_back();
    const int t = he / 3, k = he % 3;
    const int u = nbr[3 * t + k];
    if (u < 0) continue;
    int32_t* T = m.tri + 3 * t;
    int32_t* U = m.tri + 3 * u;
    const int a = T[k], b = T[(k + 1) % 3], c = T[(k + 2) % 3];  // edge (b,c), apex a in t
    // local index of the apex of u (vertex not on the shared edge)
    int ku = -1;
    for (int j = 0; j < 3; ++j)
      if (U[j] != b && U[j] != c) ku = j;
    if (ku < 0 || nbr[3 * u + ku] != t) continue;  // stale half-edge
    const int dd = U[ku];
    if (incircle(m.x + 2 * a, m.x + 2 * b, m.x + 2 * c, m.x + 2 * dd) <= 0) continue;
    // flip edge (b,c) -> (a,dd):  t = (a, b, dd),  u = (a, dd, c)   (both CCW)
    const int t_ab = nbr[3 * t + (k + 2) % 3];  // across edge (a,b) in t (opposite c)
    const int t_ca = nbr[3 * t + (k + 1) % 3];  // across edge (c,a) in t (opposite b)
    // in u: vertices cyclic (dd, c, b) since u is CCW with edge (c,b); find neighbours across (dd,c)... by vertex
    int u_bd = -1, u_dc = -1;
    for (int j = 0; j < 3; ++j) {
      if (U[j] == c) u_bd = nbr[3 * u + j];  // edge opposite c = (b,dd)
      if (U[j] == b) u_dc = nbr[3 * u + j];  // edge opposite b = (dd,c)
    }
    T[0] = a; T[1] = b; T[2] = dd;
    U[0] = a; U[1] = dd; U[2] = c;
    // t = (a,b,dd): opposite a -> edge (b,dd): u_bd ; opposite b -> edge (dd,a): u ; opposite dd -> edge (a,b): t_ab
    nbr[3 * t + 0] = u_bd; nbr[3 * t + 1] = u; nbr[3 * t + 2] = t_ab;
    // u = (a,dd,c): opposite a -> edge (dd,c): u_dc ; opposite dd -> edge (c,a): t_ca ; opposite c -> edge (a,dd): t
    nbr[3 * u + 0] = u_dc; nbr[3 * u + 1] = t_ca; nbr[3 * u + 2] = t;
    auto relink = [&](int tri_id, int old_n, int new_n) {
      if (tri_id < 0) return;
      for (int j = 0; j < 3; ++j)
        if (nbr[3 * tri_id + j] == old_n) {
          // make sure it is the right edge (a triangle can touch old_n only once in a manifold mesh)
          nbr[3 * tri_id + j] = new_n;
          return;
        }
    };
    relink(u_bd, u, t);   // edge (b,dd) now belongs to t
    relink(t_ca, t, u);   // edge (c,a) now belongs to u
    ++flips;
    for (int j = 0; j < 3; ++j) {
      if (j != 1) stack.push_back(3 * t + j);  // edges (b,dd) and (a,b)
      if (j != 2) stack.push_back(3 * u + j);  // edges (dd,c) and (c,a)
    }
  }
  return flips;
}

// DOLFIN MeshSmoothing::smooth on the triangle list (interior vertices only; every neighbour of an interior
// vertex appears in exactly two incident triangles, so the centroid is accumulated over triangles)
static void smooth(Mesh& m, const uint8_t* on_boundary, int iterations) {
  const int nv = m.nv, nt = m.nt;
  std::vector<int32_t> ptr(nv + 1, 0), inc(3 * nt);
  for (int i = 0; i < 3 * nt; ++i) ++ptr[m.tri[i] + 1];
  for (int v = 0; v < nv; ++v) ptr[v + 1] += ptr[v];
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int t = 0; t < nt; ++t)
      for (int k = 0; k < 3; ++k) inc[fill[m.tri[3 * t + k]]++] = 3 * t + k;
  }
  const double DOLFIN_EPS = 3.0e-16;
  double* x = m.x;
  for (int it = 0; it < iterations; ++it) {
    for (int v = 0; v < nv; ++v) {
      if (on_boundary[v]) continue;
      const double px = x[2 * v], py = x[2 * v + 1];
      double cx = 0.0, cy = 0.0, rmin = 0.0;
      const int n0 = ptr[v], n1 = ptr[v + 1];
      for (int q = n0; q < n1; ++q) {
        const int t = inc[q] / 3, k = inc[q] % 3;
        const int a = m.tri[3 * t + (k + 1) % 3], b = m.tri[3 * t + (k + 2) % 3];
        const double ax = x[2 * a], ay = x[2 * a + 1], bx = x[2 * b], by = x[2 * b + 1];
        cx += ax + bx;
        cy += ay + by;
        const double tx = bx - ax, ty = by - ay;
        const double nn = std::sqrt(tx * tx + ty * ty);
        const double r = std::fabs((ty * (px - ax) - tx * (py - ay)) / nn);
        rmin = (rmin == 0.0) ? r : (r < rmin ? r : rmin);
      }
      const double cnt = 2.0 * (n1 - n0);
      cx /= cnt;
      cy /= cnt;
      const double dx = cx - px, dy = cy - py;
      const double r = std::sqrt(dx * dx + dy * dy);
      if (r < DOLFIN_EPS) continue;
      const double step = (0.5 * rmin < r) ? 0.5 * rmin : r;
      x[2 * v] = px + step * dx / r;
      x[2 * v + 1] = py + step * dy / r;
    }
  }
}

static int remesh_one(double* x, int32_t* tri, int32_t* nv, int32_t* nt, int remove_idx, int smooth_iters) {
  if (remove_idx < 0) return 0;  // "do nothing" / invalid action: the reference leaves the mesh untouched
  Mesh m{*nv, *nt, x, tri};
  orient_ccw(m);
  // boundary vertices = endpoints of edges with a single owner (before the removal: the removed vertex is interior)
  // (on failure the cells go back to their canonical ascending order: orient_ccw has swapped some of them)
  auto canonical = [&] {
    for (int t = 0; t < m.nt; ++t) std::sort(m.tri + 3 * t, m.tri + 3 * t + 3);
  };
  int rc = 0;
  if (remove_idx >= 0) {
    rc = remove_vertex(m, remove_idx);
    if (rc) {
      canonical();
      return rc;
    }
  }
  const int flips = make_delaunay(m);
  if (flips < 0) {
    canonical();
    return -10 + flips;
  }
  std::vector<uint8_t> onb(m.nv, 0);
  {
    EdgeMap em(3 * m.nt);
    std::vector<int32_t> cnt;
    std::vector<std::pair<int, int>> ed;
    for (int t = 0; t < m.nt; ++t)
      for (int k = 0; k < 3; ++k) {
        const int a = m.tri[3 * t + (k + 1) % 3], b = m.tri[3 * t + (k + 2) % 3];
        bool ins;
        int32_t* s = em.find_or_insert(a, b, ins);
        if (ins) {
          *s = (int)cnt.size();
          cnt.push_back(1);
          ed.emplace_back(a, b);
        } else {
          ++cnt[*s];
        }
      }
    for (size_t e = 0; e < cnt.size(); ++e)
      if (cnt[e] == 1) onb[ed[e].first] = onb[ed[e].second] = 1;
  }
  if (smooth_iters > 0) smooth(m, onb.data(), smooth_iters);
  // canonical cells: ascending vertex ids (DOLFIN mesh.order())
  for (int t = 0; t < m.nt; ++t) std::sort(m.tri + 3 * t, m.tri + 3 * t + 3);
  *nv = m.nv;
  *nt = m.nt;
  return 0;
}

}  // namespace mdq_host

extern "C" int mdq_remesh_host(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv,
                               int32_t* nt, const int32_t* remove_idx, int32_t smooth_iters, int32_t nthreads,
                               int32_t* status) {
  if (B <= 0 || !coords || !cells || !nv || !nt || !remove_idx || !status) return mdq_set_error("mdq_remesh_host: bad arguments");
  int T = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
  if (T < 1) T = 1;
  if (T > B) T = B;
  mdq_host::Pool::get().parallel_for(B, T, [&](int b) {
    status[b] = mdq_host::remesh_one(coords + (size_t)b * NV * 2, cells + (size_t)b * NT * 3, nv + b, nt + b,
                                      remove_idx[b], smooth_iters);
  });
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Topology + N-closest selection + state graph of every environment after a remesh
// (MeshTopology / Env2DAirfoil._n_closest / get_state restated for the batched engine).
namespace mdq_host {

// squared distance from p to the segment ab (same operations as shapely's / the oracle's point-segment distance
// up to the final square root, which the caller applies once to the minimum: sqrt is monotone and correctly
// rounded, so sqrt(min d2) == min sqrt(d2) bit for bit)
static inline double seg_dist2(const double* p, const double* a, const double* b) {
  const double abx = b[0] - a[0], aby = b[1] - a[1];
  double t = ((p[0] - a[0]) * abx + (p[1] - a[1]) * aby) / (abx * abx + aby * aby);
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  const double qx = a[0] + t * abx - p[0], qy = a[1] + t * aby - p[1];
  return qx * qx + qy * qy;
}

// Segment lengths of the (fixed) airfoil polygon, once per call.
struct PolyAux {
  std::vector<double> len;
  double maxlen = 0.0;
  PolyAux(const double* poly, int np_) : len(np_) {
    for (int i = 0; i < np_; ++i) {
      const double* a = poly + 2 * i;
      const double* b = poly + 2 * ((i + 1) % np_);
      len[i] = std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]));
      maxlen = std::max(maxlen, len[i]);
    }
  }
};

static double polygon_distance(const double* poly, int np_, const PolyAux& aux, const double* p) {
  bool inside = false;
  double v2min = 1e300;  // squared distance to the nearest polygon VERTEX: an upper bound of the answer
  for (int i = 0; i < np_; ++i) {
    const double* a = poly + 2 * i;
    const double* b = poly + 2 * ((i + 1) % np_);
    if ((a[1] > p[1]) != (b[1] > p[1])) {
      const double xin = a[0] + (p[1] - a[1]) * (b[0] - a[0]) / (b[1] - a[1]);
      if (p[0] < xin) inside = !inside;
    }
    const double dx = p[0] - a[0], dy = p[1] - a[1];
    v2min = std::min(v2min, dx * dx + dy * dy);
  }
  if (inside) return 0.0;
  // a segment whose start vertex is farther than len + sqrt(bound) away cannot hold the minimum (1e-9 safety
  // margin on the squared comparison: pruning never decides between candidates that are that close)
  const double rb = std::sqrt(v2min);
  double d2 = 1e300;
  for (int i = 0; i < np_; ++i) {
    const double* a = poly + 2 * i;
    const double dx = p[0] - a[0], dy = p[1] - a[1];
    const double reach = aux.len[i] + rb;
    if (dx * dx + dy * dy > reach * reach * (1.0 + 1e-9)) continue;
    d2 = std::min(d2, seg_dist2(p, a, poly + 2 * ((i + 1) % np_)));
  }
  return std::sqrt(d2);
}

// Index data of the matrix-free IPCS path on one coarsened mesh (MeshTopology.boundary_conditions / facets /
// dof_gathers / sell_layout + IpcsBatch._host_arrays restated for the batched engine; flow_solver.py:85-132,194-226).
static int ipcs_topology_one(const mdq_env_topo_desc& D, const mdq_ipcs_topo_out& O, int b, int ne,
                             const std::vector<int32_t>& ea, const std::vector<int32_t>& eb,
                             const std::vector<int32_t>& ecnt, const std::vector<int32_t>& eown) {
  const int nv = D.nv[b], nt = D.nt[b], n2 = nv + ne;
  const double* x = D.coords + (size_t)b * D.NV * 2;
  const int32_t* tri = D.cells + (size_t)b * D.NT * 3;
  const int32_t* cd = D.cell_dofs + (size_t)b * 6 * D.NT;
  int32_t* scat = O.mf_scat + (size_t)b * 6 * D.NT;
  int8_t* cof = O.cell_outflow + (size_t)b * D.NT;
  uint8_t* fl = O.bcu_flag + (size_t)b * D.NP;
  double* gx = O.bcu_gx + (size_t)b * D.NP;
  uint8_t* pf = O.bcp_flag + (size_t)b * D.NV;
  std::fill(cof, cof + D.NT, (int8_t)-1);
  std::fill(fl, fl + D.NP, (uint8_t)0);
  std::fill(gx, gx + D.NP, 0.0);
  std::fill(pf, pf + D.NV, (uint8_t)0);
  // ---- facet tags: walls 0 / airfoil 1 / inflow 2 / outflow 3, later marks override (flow_solver.py:194-226)
  const double E = 3.0e-16;
  double bot = x[1], top = x[1];
  for (int i = 1; i < nv; ++i) {
    bot = std::min(bot, x[2 * i + 1]);
    top = std::max(top, x[2 * i + 1]);
  }
  const double H = top - bot, Um = 1.5;
  std::vector<int8_t> tag(ne, -1);
  for (int e = 0; e < ne; ++e) {
    if (ecnt[e] != 1) continue;
    const double X[3] = {x[2 * ea[e]], x[2 * eb[e]], 0.5 * (x[2 * ea[e]] + x[2 * eb[e]])};
    const double Y[3] = {x[2 * ea[e] + 1], x[2 * eb[e] + 1], 0.5 * (x[2 * ea[e] + 1] + x[2 * eb[e] + 1])};
    bool walls = true, air = true, inflow = true, outflow = true;
    for (int q = 0; q < 3; ++q) {
      walls = walls && (Y[q] > 0.5 - 2 * E || Y[q] < -0.5 + 2 * E);
      air = air && (X[q] < 3.0 - E && X[q] > -0.5 + E && Y[q] < 0.5 - E && Y[q] > -0.5 + E);
      inflow = inflow && (X[q] < -0.5 + E);
      outflow = outflow && (X[q] > 3.0 - 2 * E);
    }
    int t = 4;
    if (walls) t = 0;
    if (air) t = 1;
    if (inflow) t = 2;
    if (outflow) t = 3;
    tag[e] = (int8_t)t;
  }
  // ---- Dirichlet data, list order [inlet, airfoil, walls]: later wins on shared dofs (flow_solver.py:123-132)
  const int order[3] = {2, 1, 0};
  for (int w = 0; w < 3; ++w)
    for (int e = 0; e < ne; ++e) {
      if (tag[e] != order[w]) continue;
      const int dofs[3] = {ea[e], eb[e], nv + e};
      for (int q = 0; q < 3; ++q) {
        const int dq = dofs[q];
        fl[dq] = 1;
        if (order[w] == 2) {
          const double y = q < 2 ? x[2 * dq + 1] : 0.5 * (x[2 * ea[e] + 1] + x[2 * eb[e] + 1]);
          gx[dq] = -4.0 * Um * (y - bot) * (y - top) / H / H;
        } else {
          gx[dq] = 0.0;
        }
      }
    }
  // ---- outflow facets (edge-id order): pressure Dirichlet vertices, cell -> local facet, row list of the facet term
  struct Ent {
    int32_t row, col, src;
    bool operator<(const Ent& o) const { return row != o.row ? row < o.row : (col != o.col ? col < o.col : src < o.src); }
  };
  std::vector<Ent> ent;
  const int AL[3] = {1, 0, 0}, BL[3] = {2, 2, 1};
  for (int e = 0; e < ne; ++e) {
    if (tag[e] != 3) continue;
    pf[ea[e]] = pf[eb[e]] = 1;
    const int c = eown[e] / 3, k = eown[e] % 3;
    cof[c] = (int8_t)k;
    const int rows[3] = {AL[k], BL[k], 3 + k};
    for (int q = 0; q < 3; ++q)
      for (int j = 0; j < 6; ++j)
        ent.push_back({cd[rows[q] * D.NT + c], cd[j * D.NT + c], c * 36 + rows[q] * 6 + j});
  }
  std::sort(ent.begin(), ent.end());
  int32_t* bo_rows = O.bo_rows + (size_t)b * O.NBO;
  int32_t* bo_ptr = O.bo_ptr + (size_t)b * (O.NBO + 1);
  int32_t* bo_col = O.bo_col + (size_t)b * O.NBE;
  int32_t* bo_src = O.bo_src + (size_t)b * O.NBE;
  if ((int)ent.size() > O.NBE) return -5;
  int nbo = 0;
  int own[512];
  std::fill(own, own + 512, 0);
  bo_ptr[0] = 0;
  for (size_t t = 0; t < ent.size(); ++t) {
    if (t == 0 || ent[t].row != ent[t - 1].row) {
      if (nbo >= O.NBO) return -5;
      bo_rows[nbo++] = ent[t].row;
      if (n2 <= 4096 && ++own[ent[t].row % 512] > 2) return -4;   // (a limit of mode 3's row owners: lab-sized meshes only)
    }
    bo_col[t] = ent[t].col;
    bo_src[t] = ent[t].src;
    bo_ptr[nbo] = (int32_t)t + 1;
  }
  O.nbo[b] = nbo;
  // ---- packed per-triangle metadata of the matrix-free operators
  // (12-bit dof ids: lab-sized meshes; a larger mesh feeds mode 5, which reads cell_dofs / cell_outflow themselves)
  if (n2 <= 4096)
    for (int i = 0; i < 6; ++i)
      for (int t = 0; t < nt; ++t) scat[i * D.NT + t] = cd[i * D.NT + t] | (i == 0 ? ((int32_t)(cof[t] + 1) << 28) : 0);
  // ---- dof <- element-slot gathers (ascending slots)
  auto gather = [&](int nl, int ndof, int32_t* ptr, int32_t* src) {
    std::fill(ptr, ptr + ndof + 1, 0);
    for (int t = 0; t < nt; ++t)
      for (int i = 0; i < nl; ++i) ++ptr[cd[i * D.NT + t] + 1];
    for (int i = 0; i < ndof; ++i) ptr[i + 1] += ptr[i];
    std::vector<int32_t> fill(ptr, ptr + ndof);
    for (int t = 0; t < nt; ++t)
      for (int i = 0; i < nl; ++i) src[fill[cd[i * D.NT + t]]++] = t * nl + i;
  };
  int32_t* g1p = O.g1_ptr + (size_t)b * (D.NV + 1);
  gather(3, nv, g1p, O.g1_src + (size_t)b * 3 * D.NT);
  gather(6, n2, O.g2_ptr + (size_t)b * (D.NP + 1), O.g2_src + (size_t)b * 6 * D.NT);
  // ---- SELL-64 pattern of the P1 Laplacian: row = {vertex} + neighbours, ascending; slice width = longest row
  std::vector<int32_t> nptr(nv + 1, 0);
  for (int e = 0; e < ne; ++e) {
    ++nptr[ea[e] + 1];
    ++nptr[eb[e] + 1];
  }
  for (int i = 0; i < nv; ++i) nptr[i + 1] += nptr[i] + 1;  // +1: the diagonal
  std::vector<int32_t> cols(nptr[nv]), fillp(nptr.begin(), nptr.end() - 1);
  for (int i = 0; i < nv; ++i) cols[fillp[i]++] = i;
  for (int e = 0; e < ne; ++e) {
    cols[fillp[ea[e]]++] = eb[e];
    cols[fillp[eb[e]]++] = ea[e];
  }
  for (int i = 0; i < nv; ++i) std::sort(cols.begin() + nptr[i], cols.begin() + nptr[i + 1]);
  int32_t* so = O.sl1_off + (size_t)b * (D.NV / 64 + 2);
  int32_t* sc = O.sl1_col + (size_t)b * O.NSE1;
  const int ns = (nv + 63) / 64;
  so[0] = 0;
  for (int s_ = 0; s_ < ns; ++s_) {
    int w = 0;
    for (int r = 64 * s_; r < std::min(nv, 64 * s_ + 64); ++r) w = std::max(w, nptr[r + 1] - nptr[r]);
    so[s_ + 1] = so[s_] + 64 * w;
    if (so[s_ + 1] > O.NSE1) return -6;
    for (int l = 0; l < 64; ++l) {
      const int r = 64 * s_ + l, rr = std::min(r, nv - 1);
      const int len = r < nv ? nptr[r + 1] - nptr[r] : 0;
      for (int j = 0; j < w; ++j) sc[so[s_] + 64 * j + l] = j < len ? cols[nptr[r] + j] : rr;  // padding: own row, value 0
    }
  }
  return 0;
}

static int topology_one(const mdq_env_topo_desc& D, int b) {
  const int nv = D.nv[b], nt = D.nt[b];
  const double* x = D.coords + (size_t)b * D.NV * 2;
  const int32_t* tri = D.cells + (size_t)b * D.NT * 3;
  int32_t* cd = D.cell_dofs + (size_t)b * 6 * D.NT;
  double* pts = D.points + (size_t)b * D.NP * 2;
  // ---- edges numbered by first appearance in (cell, local edge k opposite vertex k) order
  EdgeMap em(3 * nt);
  std::vector<int32_t> ea, eb, ecnt, eown;
  ea.reserve(3 * nt / 2 + 64);
  eb.reserve(3 * nt / 2 + 64);
  for (int t = 0; t < nt; ++t) {
    const int32_t* v = tri + 3 * t;
    for (int k = 0; k < 3; ++k) {
      const int a = v[k == 0 ? 1 : 0], bb = v[k == 2 ? 1 : 2];
      bool ins;
      int32_t* s = em.find_or_insert(a, bb, ins);
      if (ins) {
        *s = (int)ea.size();
        ea.push_back(a);
        eb.push_back(bb);
        ecnt.push_back(1);
        eown.push_back(3 * t + k);
      } else {
        ++ecnt[*s];
      }
      cd[(3 + k) * D.NT + t] = nv + *s;
      cd[k * D.NT + t] = v[k];
    }
  }
  const int ne = (int)ea.size();
  if (nv + ne > D.NP) return -1;
  D.ne[b] = ne;
  for (int i = 0; i < nv; ++i) {
    pts[2 * i] = x[2 * i];
    pts[2 * i + 1] = x[2 * i + 1];
  }
  for (int e = 0; e < ne; ++e) {
    pts[2 * (nv + e)] = 0.5 * (x[2 * ea[e]] + x[2 * eb[e]]);
    pts[2 * (nv + e) + 1] = 0.5 * (x[2 * ea[e] + 1] + x[2 * eb[e] + 1]);
  }
  // ---- boundary vertices, airfoil facets (tag 1: all of both end points + midpoint strictly inside the box)
  std::vector<uint8_t> onb(nv, 0);
  int naf = 0;
  int32_t* af = D.af_facets + (size_t)b * D.NAF * 2;
  const double E = 3.0e-16;
  auto in_air = [&](double px, double py) { return px < 3.0 - E && px > -0.5 + E && py < 0.5 - E && py > -0.5 + E; };
  for (int e = 0; e < ne; ++e) {
    if (ecnt[e] != 1) continue;
    onb[ea[e]] = onb[eb[e]] = 1;
    const double ax = x[2 * ea[e]], ay = x[2 * ea[e] + 1], bx = x[2 * eb[e]], by = x[2 * eb[e] + 1];
    if (in_air(ax, ay) && in_air(bx, by) && in_air(0.5 * (ax + bx), 0.5 * (ay + by))) {
      if (naf >= D.NAF) return -2;
      af[2 * naf] = eown[e] / 3;
      af[2 * naf + 1] = eown[e] % 3;
      ++naf;
    }
  }
  D.naf[b] = naf;
  // ---- removable: `coord not in bmesh.coordinates()` = neither x nor y equals ANY boundary x / y (numpy quirk)
  std::vector<double> bx, by;
  for (int i = 0; i < nv; ++i)
    if (onb[i]) {
      bx.push_back(x[2 * i]);
      by.push_back(x[2 * i + 1]);
    }
  std::sort(bx.begin(), bx.end());
  std::sort(by.begin(), by.end());
  std::vector<int32_t> removable;
  removable.reserve(nv);
  for (int i = 0; i < nv; ++i) {
    const bool hit = std::binary_search(bx.begin(), bx.end(), x[2 * i]) || std::binary_search(by.begin(), by.end(), x[2 * i + 1]);
    if (!hit) removable.push_back(i);
  }
  const int nrem = (int)removable.size();
  D.nremovable[b] = nrem;
  // ---- N closest removable vertices to the airfoil polygon (argsort of the distances, window by offset)
  std::vector<double> dist(nrem);
  const PolyAux aux(D.polygon, D.npoly);
  for (int r = 0; r < nrem; ++r) dist[r] = polygon_distance(D.polygon, D.npoly, aux, x + 2 * removable[r]);
  std::vector<int32_t> order(nrem);
  for (int r = 0; r < nrem; ++r) order[r] = r;
  std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return dist[a] < dist[c]; });
  const int off = D.offset[b];
  int nsel = nrem - off;
  if (nsel > D.N) nsel = D.N;
  if (nsel < 0) nsel = 0;
  D.nsel[b] = nsel;
  int32_t* nc = D.n_closest + (size_t)b * D.N;
  int32_t* cm = D.coord_map + (size_t)b * D.N;
  std::vector<int32_t> inv(nv, -1);
  for (int i = 0; i < nsel; ++i) {
    nc[i] = order[off + i];
    cm[i] = removable[order[off + i]];
    inv[cm[i]] = i;
  }
  for (int i = nsel; i < D.N; ++i) nc[i] = cm[i] = 0;
  // ---- state graph: cells whose three vertices are all selected -> edges (id1,id2),(id1,id3),(id2,id3)
  int32_t* es = D.edge_src + (size_t)b * D.EMAX;
  int32_t* ed = D.edge_dst + (size_t)b * D.EMAX;
  double* el = D.edge_len + (size_t)b * D.EMAX;
  int E_ = 0;
  for (int t = 0; t < nt; ++t) {
    const int32_t* v = tri + 3 * t;
    const int i0 = inv[v[0]], i1 = inv[v[1]], i2 = inv[v[2]];
    if (i0 < 0 || i1 < 0 || i2 < 0) continue;
    if (E_ + 3 > D.EMAX) return -3;
    const int pa[3] = {0, 0, 1}, pb[3] = {1, 2, 2}, id[3] = {i0, i1, i2};
    for (int q = 0; q < 3; ++q) {
      es[E_] = id[pa[q]];
      ed[E_] = id[pb[q]];
      const double dx = x[2 * v[pa[q]]] - x[2 * v[pb[q]]], dy = x[2 * v[pa[q]] + 1] - x[2 * v[pb[q]] + 1];
      el[E_] = std::sqrt(dx * dx + dy * dy);
      ++E_;
    }
  }
  D.nedges[b] = E_;
  if (D.ipcs) return ipcs_topology_one(D, *D.ipcs, b, ne, ea, eb, ecnt, eown);
  return 0;
}

}  // namespace mdq_host

extern "C" int mdq_env_topology_host(const mdq_env_topo_desc* d, int32_t nthreads, int32_t* status) {
  if (!d || d->B <= 0 || !status) return mdq_set_error("mdq_env_topology_host: bad arguments");
  const mdq_env_topo_desc D = *d;
  int T = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
  if (T < 1) T = 1;
  if (T > D.B) T = D.B;
  mdq_host::Pool::get().parallel_for(D.B, T, [&](int b) { status[b] = mdq_host::topology_one(D, b); });
  return 0;
}
