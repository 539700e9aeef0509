// Device version of mdq_remesh_host without the smoothing (which is mdq_smooth): Env2DAirfoil._remove_vertex
// (Env2DAirfoil.py:452-512) for B meshes, one 256-thread workgroup per mesh out of LDS.
//
// The reference deletes the vertex and re-runs a GLOBAL scipy/Qhull Delaunay of the remaining points, then drops the
// simplices whose vertices are all boundary vertices.  Because the boundary never changes, that equals: re-triangulate
// the cavity of the removed vertex (ear clipping of its star polygon) and restore the Delaunay property of the whole
// triangulation with Lawson flips (the smoothing of the previous step has moved every vertex, so edges anywhere may
// have to flip; measured 0.3-0.4 triangles per step outside the cavity).  The Delaunay triangulation is unique, so
// the result is the reference's as a SET of cells (verified against scipy for the host engine, which this kernel
// reproduces slot for slot).
//
// Parallel: orientation, star search, vertex renumbering, neighbour table (LDS hash), detection of the edges that
// violate the empty-circle test, final cell ordering.  Serial (lane 0, a few thousand instructions): ear clipping of
// the <= 64-gon and the flip work list.  Predicates are the host engine's, operation by operation (no FMA).
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_rm {

#ifndef MDQ_REMESH_WG
#define MDQ_REMESH_WG 1024
#endif
constexpr int RW = MDQ_REMESH_WG;   // threads per mesh (256 in rounds 1-3: the parallel phases are loops of LDS atomics / hash probes, i.e. latency chains)
// Capacities as a template parameter K (round 4): K = 1 - 1024 vertices / 2048 triangles, every table in LDS (the
// kernel of the 128-environment batches); K = 4 - 4096 vertices / 8192 triangles (BASELINE configs[4]: ys930 red-refined,
// 3 322 vertices), the same code on a slab in GLOBAL memory per mesh (L2 resident; slower: every table access is an L2
// round trip instead of an LDS one - the reference's _remove_vertex takes whatever mesh it is given, Env2DAirfoil.py:452-512);
// K = 16 (round 6) - 16 384 vertices / 32 768 triangles (ys930 red-refined twice: 12 924 / 25 120): the slab form throughout, the
// edge hash of 131 072 slots on the slab as well (it does not fit the LDS) - coverage, not speed.
template <int K>
struct Cap {
  static constexpr int NV = 1024 * K, NT = 2048 * K, NS = 3 * NT, HS = 8192 * K;
  static constexpr int VBITS = K == 1 ? 10 : (K == 4 ? 12 : 14), HSHIFT = K == 1 ? 19 : (K == 4 ? 17 : 15);   // vertex ids / 32 - log2(HS)
  static constexpr size_t BYTES = (size_t)16 * NV + 2 * sizeof(int) * NS + 2 * sizeof(uint32_t) * HS + sizeof(int) * 336;
};
constexpr int RNV = Cap<1>::NV, RNT = Cap<1>::NT;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ double orient2d(double2 a, double2 b, double2 c) {
#pragma clang fp contract(off)
  return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
}
// > 0 when d lies inside the circumcircle of the CCW triangle (a,b,c)
__device__ __forceinline__ double incircle(double2 a, double2 b, double2 c, double2 d) {
#pragma clang fp contract(off)
  const double ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
  return (ax * ax + ay * ay) * (bx * cy - by * cx) - (bx * bx + by * by) * (ax * cy - ay * cx) +
         (cx * cx + cy * cy) * (ax * by - ay * bx);
}
template <int SH>
__device__ __forceinline__ uint32_t hslot(uint32_t key) { return (key * 2654435761u) >> SH; }

// the action decoding of mdq_env_act (env_act_kernel, mdq_mesh.hip) as the head of the removal kernel
struct ActArgs {
  int N;
  const float* q;
  const uint8_t* explore;
  const int32_t* rand_action;
  const int32_t* nsel;
  const int32_t* coord_map;
  int32_t* offset;
  int32_t* action;
  int32_t* rem;
  int32_t* code;
};

#ifdef MDQ_RM_TRACE
// debug build only: s_memtime deltas of thread 0 of mesh 0 at the phase boundaries
__device__ long long mdq_rm_trace_buf[16];
#define RM_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_rm_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
extern "C" int mdq_rm_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_rm_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_rm_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define RM_STAMP(k)
#endif

template <bool ACT, int K>
__global__ __launch_bounds__(RW) void remesh_kernel(int NV, int NT, double* coords, int32_t* cells, int32_t* nv_,
                                                    int32_t* nt_, const int32_t* remove_idx, int32_t* status, ActArgs A,
                                                    unsigned char* slab) {
#pragma clang fp contract(off)
  using C = Cap<K>;
  constexpr int RNV = C::NV, RNS = C::NS, RHS = C::HS;
  extern __shared__ __align__(16) unsigned char lds_[];
  unsigned char* smem = K == 1 ? lds_ : slab + (size_t)blockIdx.x * ((C::BYTES + 255) & ~(size_t)255);
  double2* X = reinterpret_cast<double2*>(smem);                     // [RNV]            16 KB (K = 1)
  int* tri = reinterpret_cast<int*>(smem + 16 * RNV);                // [RNT][3]         24 KB
  int* nbr = tri + RNS;                                              // [RNT][3]         24 KB
  // K = 4 (round 5): the hash of the neighbour table - and the flip stack that re-uses it - in LDS, 16 384 slots (load <= 0.75
  // at 12 288 edges; probes bounded): its inserts were atomics + dependent reads on the slab, an L2 round trip each, 75 % of the
  // large-mesh instance (tools/trace_remesh.py on the refined ys930: 557 k of 743 k cycles).  Everything else stays on the slab.
  constexpr int HT = K == 4 ? 16384 : RHS;
  constexpr int HSH = K == 4 ? 18 : C::HSHIFT;
  // (K = 1: behind the neighbour table in LDS; K = 4: the LDS; K = 16: behind the neighbour table on the slab)
  uint32_t* hkey = K == 4 ? reinterpret_cast<uint32_t*>(lds_) : reinterpret_cast<uint32_t*>(nbr + RNS);   // [HT] | the flip stack
  uint32_t* hval = hkey + HT;                                        // [HT]             | re-uses this region
  int* stack = reinterpret_cast<int*>(hkey);                         // [2 * HT]
  int* misc = reinterpret_cast<int*>(reinterpret_cast<uint32_t*>(nbr + RNS) + 2 * RHS);   // [336]: counters, star list, lane 0's work arrays
  int* star = misc + 8;                                              // [64]
  int* lane0 = misc + 80;                                            // [4][64] ring bookkeeping of the serial part
  const int b = blockIdx.x, tid = threadIdx.x;
#ifdef MDQ_RM_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  int rv;
  if (ACT) {
    // wave 0: greedy action = first maximum of the Q-row (torch.argmax), epsilon-greedy choice, Env2DAirfoil.step's decoding
    if (tid < 64) {
      const int lane = tid, N = A.N, NA = N + 1;
      int a;
      if (A.q) {
        float best = -__builtin_inff();
        int bi = 0x7FFFFFFF;
        for (int i = lane; i < NA; i += 64) {
          const float v = A.q[(int64_t)b * NA + i];
          if (v > best || (v == best && i < bi) || (bi == 0x7FFFFFFF && !(v < best))) {   // (NaN rows: first index)
            best = v;
            bi = i;
          }
        }
        for (int off = 32; off > 0; off >>= 1) {
          const float ov = __shfl_xor(best, off);
          const int oi = __shfl_xor(bi, off);
          if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
          }
        }
        a = (A.explore && A.explore[b]) ? A.rand_action[b] : bi;
      } else {
        a = A.action[b];
      }
      if (lane == 0) {
        const bool shift = a == N;
        const bool pick = a >= 0 && a < A.nsel[b] && !shift;
        if (shift) A.offset[b] += 1;
        A.action[b] = a;
        const int r = pick ? A.coord_map[(int64_t)b * N + min(max(a, 0), N - 1)] : -1;
        A.rem[b] = r;
        A.code[b] = (!shift && !pick) ? 2 : 0;
        misc[2] = r;
      }
    }
    __syncthreads();
    rv = misc[2];
    __syncthreads();
  } else {
    rv = remove_idx[b];
  }
  RM_STAMP(0)
  if (tid == 0) status[b] = 0;
  if (rv < 0) return;  // "do nothing" / invalid action: the reference leaves the mesh untouched
  int nv = nv_[b], nt = nt_[b];
  double2* xg = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  int32_t* tg = cells + (int64_t)b * NT * 3;
  for (int v = tid; v < nv; v += RW) X[v] = xg[v];
  if (tid == 0) misc[0] = 0;
  __syncthreads();
  // counter-clockwise cells
  for (int t = tid; t < nt; t += RW) {
    int v0 = tg[3 * t], v1 = tg[3 * t + 1], v2 = tg[3 * t + 2];
    if (orient2d(X[v0], X[v1], X[v2]) < 0) {
      const int w = v1;
      v1 = v2;
      v2 = w;
    }
    tri[3 * t] = v0;
    tri[3 * t + 1] = v1;
    tri[3 * t + 2] = v2;
    if (v0 == rv || v1 == rv || v2 == rv) {
      const int q = atomicAdd(&misc[0], 1);
      if (q < 64) star[q] = t;
    }
  }
  __syncthreads();
  RM_STAMP(1)
  const int ns = misc[0];
  // ---------------- lane 0: ring of the star, ear clipping, slot bookkeeping (host remove_vertex, slot for slot)
  if (tid == 0) {
    int rc = 0;
    // (in LDS: as local arrays with run-time indices they lived in scratch memory, a global round trip per access)
    int *ea = lane0, *eb = lane0 + 64, *ring = lane0 + 128, *poly = lane0 + 192;
    if (ns > 64) rc = -1;
    else if (ns < 3) rc = -2;
    if (rc == 0) {
      for (int i = 1; i < ns; ++i) {  // ascending cell ids, like the host's scan
        const int w = star[i];
        int j = i - 1;
        while (j >= 0 && star[j] > w) {
          star[j + 1] = star[j];
          --j;
        }
        star[j + 1] = w;
      }
      for (int s = 0; s < ns; ++s) {
        const int* v = tri + 3 * star[s];
        const int k = v[0] == rv ? 0 : (v[1] == rv ? 1 : 2);
        ea[s] = v[(k + 1) % 3];
        eb[s] = v[(k + 2) % 3];
      }
      ring[0] = ea[0];
      int cur = eb[0];
      for (int n = 1; n < ns && rc == 0; ++n) {
        ring[n] = cur;
        int f = -1;
        for (int s = 0; s < ns; ++s)
          if (ea[s] == cur) f = s;
        if (f < 0) rc = -3;  // open star: rv is a boundary vertex
        else cur = eb[f];
      }
      if (rc == 0 && cur != ring[0]) rc = -3;
    }
    int nn = 0;
    if (rc == 0) {
      int np_ = ns;
      for (int i = 0; i < ns; ++i) poly[i] = ring[i];
      int* newtri = stack;  // (the stack region is free until the flip phase)
      int guard = 0;
      while (np_ > 3 && guard++ < 4096 && rc == 0) {
        bool clipped = false;
        for (int i = 0; i < np_ && !clipped; ++i) {
          const int p0 = poly[(i + np_ - 1) % np_], p1 = poly[i], p2 = poly[(i + 1) % np_];
          const double2 A = X[p0], Bp = X[p1], Cp = X[p2];
          if (orient2d(A, Bp, Cp) <= 0) continue;  // reflex corner
          bool empty = true;
          for (int j = 0; j < np_ && empty; ++j) {
            const int q = poly[j];
            if (q == p0 || q == p1 || q == p2) continue;
            const double2 Q = X[q];
            if (orient2d(A, Bp, Q) >= 0 && orient2d(Bp, Cp, Q) >= 0 && orient2d(Cp, A, Q) >= 0) empty = false;
          }
          if (!empty) continue;
          newtri[3 * nn] = p0;
          newtri[3 * nn + 1] = p1;
          newtri[3 * nn + 2] = p2;
          ++nn;
          for (int j = i; j + 1 < np_; ++j) poly[j] = poly[j + 1];
          --np_;
          clipped = true;
        }
        if (!clipped) rc = -4;
      }
      if (rc == 0 && np_ != 3) rc = -4;
      if (rc == 0) {
        newtri[3 * nn] = poly[0];
        newtri[3 * nn + 1] = poly[1];
        newtri[3 * nn + 2] = poly[2];
        ++nn;  // nn == ns - 2
        for (int s = 0; s < nn; ++s)
          for (int k = 0; k < 3; ++k) tri[3 * star[s] + k] = newtri[3 * s + k];
        int dead0 = star[ns - 2], dead1 = star[ns - 1];
        if (dead0 < dead1) {
          const int w = dead0;
          dead0 = dead1;
          dead1 = w;
        }
        const int dead[2] = {dead0, dead1};  // the higher slot first
        for (int q = 0; q < 2; ++q) {
          const int last = nt - 1;
          if (dead[q] != last)
            for (int k = 0; k < 3; ++k) tri[3 * dead[q] + k] = tri[3 * last + k];
          --nt;
        }
      }
    }
    misc[1] = rc;
    misc[2] = nt;
  }
  __syncthreads();
  if (misc[1] != 0) {
    if (tid == 0) status[b] = misc[1];
    return;  // mesh untouched (nothing has been written back)
  }
  RM_STAMP(2)
  nt = misc[2];
  // drop the vertex: ids above shift down, coordinates move up
  for (int i = tid; i < 3 * nt; i += RW)
    if (tri[i] > rv) --tri[i];
  double2 moved[RNV / RW];
#pragma unroll
  for (int i = 0; i < RNV / RW; ++i) {
    const int v = tid + i * RW;
    moved[i] = (v >= rv && v + 1 < nv) ? X[v + 1] : make_double2(0.0, 0.0);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RNV / RW; ++i) {
    const int v = tid + i * RW;
    if (v >= rv && v + 1 < nv) X[v] = moved[i];
  }
  --nv;
  RM_STAMP(3)
  // ---------------- neighbour table: nbr[3t+k] = cell across the edge opposite local vertex k
  for (int h = tid; h < HT; h += RW) {
    hkey[h] = EMPTY;
    hval[h] = EMPTY;
  }
  for (int s = tid; s < 3 * nt; s += RW) nbr[s] = -1;
  if (tid == 0) {
    misc[3] = 0;  // non-manifold flag
    misc[4] = 0;  // stack size
  }
  __syncthreads();
  for (int s = tid; s < 3 * nt; s += RW) {
    const int t = s / 3, k = s - 3 * t;
    const int a = tri[3 * t + (k + 1) % 3], c = tri[3 * t + (k + 2) % 3];
    const uint32_t key = ((uint32_t)min(a, c) << C::VBITS) | (uint32_t)max(a, c);
    uint32_t h = hslot<HSH>(key);
    int n_ = 0;
    for (;; ++n_) {
      const uint32_t old = atomicCAS(&hkey[h], EMPTY, key);
      if (old == EMPTY || old == key || n_ >= HT) break;
      h = (h + 1) & (HT - 1);
    }
    if (n_ >= HT) {                 // more distinct edges than slots: not a triangulation within the capacities
      misc[3] = 1;
      continue;
    }
    const uint32_t o = atomicCAS(&hval[h], EMPTY, (uint32_t)s);
    if (o != EMPTY) {  // the second owner links both half-edges
      if (atomicCAS(reinterpret_cast<unsigned int*>(&nbr[o]), 0xFFFFFFFFu, (unsigned int)t) != 0xFFFFFFFFu) misc[3] = 1;
      nbr[s] = (int)(o / 3);
    }
  }
  __syncthreads();
  RM_STAMP(4)
  if (misc[3]) {
    if (tid == 0) status[b] = -11;  // non-manifold (-10 + make_delaunay's -1)
    return;
  }
  // ---------------- edges that violate the empty-circle test (each interior edge once, from its lower cell)
  // (the hash is dead: its memory becomes the work list; the barrier above separates the two uses)
  for (int s = tid; s < 3 * nt; s += RW) {
    const int t = s / 3, k = s - 3 * t, u = nbr[s];
    if (u > t) {
      const int a = tri[3 * t + k], bb = tri[3 * t + (k + 1) % 3], c = tri[3 * t + (k + 2) % 3];
      int dd = -1;
      for (int j = 0; j < 3; ++j) {
        const int w = tri[3 * u + j];
        if (w != bb && w != c) dd = w;
      }
      if (dd >= 0 && incircle(X[a], X[bb], X[c], X[dd]) > 0) stack[atomicAdd(&misc[4], 1)] = s;
    }
  }
  __syncthreads();
  RM_STAMP(5)
  // ---------------- lane 0: Lawson flips from the violating edges (host make_delaunay's loop)
  if (tid == 0) {
    int sp = misc[4];
    for (int i = 1; i < sp; ++i) {  // deterministic order of the initial list
      const int w = stack[i];
      int j = i - 1;
      while (j >= 0 && stack[j] > w) {
        stack[j + 1] = stack[j];
        --j;
      }
      stack[j + 1] = w;
    }
    int rc = 0, guard = 0;
    while (sp > 0) {
      if (++guard > 200000 || sp + 4 > 2 * HT) {
        rc = -12;
        break;
      }
      const int he = stack[--sp];
      const int t = he / 3, k = he % 3;
      const int u = nbr[3 * t + k];
      if (u < 0) continue;
      int* T = tri + 3 * t;
      int* U = tri + 3 * u;
      const int a = T[k], bb = T[(k + 1) % 3], c = T[(k + 2) % 3];  // edge (bb,c), apex a in t
      int ku = -1;
      for (int j = 0; j < 3; ++j)
        if (U[j] != bb && U[j] != c) ku = j;
      if (ku < 0 || nbr[3 * u + ku] != t) continue;  // stale half-edge
      const int dd = U[ku];
      if (incircle(X[a], X[bb], X[c], X[dd]) <= 0) continue;
      const int t_ab = nbr[3 * t + (k + 2) % 3];
      const int t_ca = nbr[3 * t + (k + 1) % 3];
      int u_bd = -1, u_dc = -1;
      for (int j = 0; j < 3; ++j) {
        if (U[j] == c) u_bd = nbr[3 * u + j];
        if (U[j] == bb) u_dc = nbr[3 * u + j];
      }
      T[0] = a; T[1] = bb; T[2] = dd;
      U[0] = a; U[1] = dd; U[2] = c;
      nbr[3 * t + 0] = u_bd; nbr[3 * t + 1] = u; nbr[3 * t + 2] = t_ab;
      nbr[3 * u + 0] = u_dc; nbr[3 * u + 1] = t_ca; nbr[3 * u + 2] = t;
      if (u_bd >= 0)
        for (int j = 0; j < 3; ++j)
          if (nbr[3 * u_bd + j] == u) {
            nbr[3 * u_bd + j] = t;
            break;
          }
      if (t_ca >= 0)
        for (int j = 0; j < 3; ++j)
          if (nbr[3 * t_ca + j] == t) {
            nbr[3 * t_ca + j] = u;
            break;
          }
      stack[sp++] = 3 * t + 0;
      stack[sp++] = 3 * t + 2;
      stack[sp++] = 3 * u + 0;
      stack[sp++] = 3 * u + 1;
    }
    misc[1] = rc;
  }
  __syncthreads();
  RM_STAMP(6)
  if (misc[1] != 0) {
    if (tid == 0) status[b] = misc[1];
    return;
  }
  // ---------------- canonical cells (ascending vertex ids, DOLFIN mesh.order()) and write-back
  for (int t = tid; t < nt; t += RW) {
    int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2], w;
    if (v0 > v1) { w = v0; v0 = v1; v1 = w; }
    if (v1 > v2) { w = v1; v1 = v2; v2 = w; }
    if (v0 > v1) { w = v0; v0 = v1; v1 = w; }
    tg[3 * t] = v0;
    tg[3 * t + 1] = v1;
    tg[3 * t + 2] = v2;
  }
  for (int v = tid; v < nv; v += RW) xg[v] = X[v];
  if (tid == 0) {
    nv_[b] = nv;
    nt_[b] = nt;
  }
  RM_STAMP(7)
}

}  // namespace mdq_rm

template <int K>
static int remesh_launch_k(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv, int32_t* nt,
                           const int32_t* remove_idx, int32_t* status, const mdq_rm::ActArgs* act, void* stream, unsigned char* slab) {
  const size_t lds = K == 1 ? mdq_rm::Cap<1>::BYTES : (K == 4 ? 2 * sizeof(uint32_t) * 16384 : 0);   // (K = 4: the edge hash / flip stack)
  static const hipError_t attr = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_rm::remesh_kernel<false, K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_rm::remesh_kernel<true, K>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
  }();
  if (attr != hipSuccess) return mdq_set_error("hipFuncSetAttribute(remesh_kernel) failed");
  if (act)
    hipLaunchKernelGGL((mdq_rm::remesh_kernel<true, K>), dim3(B), dim3(mdq_rm::RW), lds, (hipStream_t)stream, NV, NT, coords,
                       cells, nv, nt, remove_idx, status, *act, slab);
  else
    hipLaunchKernelGGL((mdq_rm::remesh_kernel<false, K>), dim3(B), dim3(mdq_rm::RW), lds, (hipStream_t)stream, NV, NT, coords,
                       cells, nv, nt, remove_idx, status, mdq_rm::ActArgs{}, slab);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("remesh_kernel launch failed");
  return 0;
}

extern "C" int64_t mdq_remesh_workspace_bytes(int32_t B, int32_t NV, int32_t NT) {
  if (B <= 0 || NV <= 0 || NT <= 0) return 0;
  if (NV <= mdq_rm::Cap<1>::NV && NT <= mdq_rm::Cap<1>::NT) return 0;         // every table in LDS
  if (NV > mdq_rm::Cap<16>::NV || NT > mdq_rm::Cap<16>::NT) return -1;         // beyond the kernels
  if (NV > mdq_rm::Cap<4>::NV || NT > mdq_rm::Cap<4>::NT) return (int64_t)((mdq_rm::Cap<16>::BYTES + 255) & ~(size_t)255) * B;
  return (int64_t)((mdq_rm::Cap<4>::BYTES + 255) & ~(size_t)255) * B;
}

static int remesh_launch(const char* who, int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv,
                         int32_t* nt, const int32_t* remove_idx, int32_t* status, const mdq_rm::ActArgs* act, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  if (B <= 0 || !coords || !cells || !nv || !nt || (!remove_idx && !act) || !status) return mdq_set_error("mdq_remesh: bad arguments");
  (void)who;
  if (NV <= mdq_rm::Cap<1>::NV && NT <= mdq_rm::Cap<1>::NT)
    return remesh_launch_k<1>(B, NV, NT, coords, cells, nv, nt, remove_idx, status, act, stream, nullptr);   // (no workspace needed)
  if (NV > mdq_rm::Cap<16>::NV || NT > mdq_rm::Cap<16>::NT)
    return mdq_set_error("mdq_remesh: capacity above 16384 vertices / 32768 triangles");
  // the large-mesh instance: its tables live in the CALLER's workspace (global memory, one slab per mesh)
  if (!workspace || workspace_bytes < mdq_remesh_workspace_bytes(B, NV, NT) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return mdq_set_error("mdq_remesh: workspace missing, too small or not 16-byte aligned (mdq_remesh_workspace_bytes)");
  if (NV > mdq_rm::Cap<4>::NV || NT > mdq_rm::Cap<4>::NT)
    return remesh_launch_k<16>(B, NV, NT, coords, cells, nv, nt, remove_idx, status, act, stream, static_cast<unsigned char*>(workspace));
  return remesh_launch_k<4>(B, NV, NT, coords, cells, nv, nt, remove_idx, status, act, stream, static_cast<unsigned char*>(workspace));
}

extern "C" int mdq_remesh(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv, int32_t* nt,
                          const int32_t* remove_idx, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!remove_idx) return mdq_set_error("mdq_remesh: bad arguments");
  return remesh_launch("mdq_remesh", B, NV, NT, coords, cells, nv, nt, remove_idx, status, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mdq_remesh_act(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv, int32_t* nt,
                              int32_t N, const float* q, const uint8_t* explore, const int32_t* rand_action,
                              const int32_t* nsel, const int32_t* coord_map, int32_t* offset, int32_t* action, int32_t* rem,
                              int32_t* code, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream) {
  if (N <= 0 || !nsel || !coord_map || !offset || !action || !rem || !code || (explore && !rand_action))
    return mdq_set_error("mdq_remesh_act: bad arguments");
  const mdq_rm::ActArgs a{N, q, explore, rand_action, nsel, coord_map, offset, action, rem, code};
  return remesh_launch("mdq_remesh_act", B, NV, NT, coords, cells, nv, nt, nullptr, status, &a, workspace, workspace_bytes, stream);
}
