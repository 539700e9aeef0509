// DOLFIN `Mesh.smooth(n)` (flow_solver.py:65-67, 236-237) for meshes BEYOND the 1024-vertex kernels (mdq_smooth.hip,
// mdq_smooth_linear.hip): up to 4096 vertices / 8192 triangles, one 512-thread workgroup per mesh (round 4: the reference
// smooths whatever mesh it is given - BASELINE configs[4] is ys930 red-refined, 3 322 vertices).
//
// A Gauss-Seidel sweep in vertex order is a dependency DAG: an interior vertex needs the NEW positions of its lower-numbered
// interior neighbours and the OLD ones of everything else.  level(v) = 1 + max level of its lower-numbered interior
// neighbours (0 for fixed vertices): two neighbours never share a level, so all vertices of a level can take DOLFIN's exact
// update (centroid of the neighbours, step limited to half the smallest altitude over the incident cells' opposite edges,
// "stays" below DOLFIN_EPS) at once and IN PLACE - exact sequential semantics, a workgroup barrier per level.  Positions live
// in LDS (64 KB), the vertex -> cells table (sorted by cell: the fixed summation order of the other kernels' exact update)
// and the level schedule on a slab in global memory.  Straightforward, not fast: ~1.5 ms for 50 sweeps of the refined ys930.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_smooth_big {
constexpr int BNV = 4096, BNT = 8192, BWG = 1024;
constexpr int MAXK = 16;           // incident cells per vertex in the level-ordered work records (more: the general loop)
constexpr int MAXLEV = 1024;       // levels whose starts are kept in LDS (more: the general loop)
constexpr size_t SLAB_BYTES = sizeof(int) * (BNV + 8) + sizeof(uint32_t) * 3 * BNT * 2 + sizeof(uint16_t) * BNV * 2 + sizeof(int) * (BNV + 8) +
                              sizeof(uint32_t) * BNV + sizeof(uint32_t) * BNV * MAXK;
// KV = 4 (round 6): 16 384 vertices / 32 768 triangles (ys930 red-refined twice: 12 924 / 25 120) - the level kernel with the
// positions and the level array on the slab too (they do not fit the LDS): every update gathers from L2 and every level ends in a
// full barrier.  Coverage, not speed: the reference's Mesh.smooth takes whatever mesh it is given (flow_solver.py:236-237).
template <int KV>
struct BCap {
  static constexpr int NV = 4096 * KV, NT = 8192 * KV;
  static constexpr size_t TABLES = sizeof(int) * (NV + 8) + sizeof(uint32_t) * 3 * NT * 2 + sizeof(uint16_t) * NV * 2 + sizeof(int) * (NV + 8) +
                                   sizeof(uint32_t) * NV + sizeof(uint32_t) * NV * MAXK;
  static constexpr size_t SLAB = TABLES + (KV > 1 ? 16 * (size_t)NV + sizeof(int) * NV + 64 : 0);     // + positions + levels
};
static_assert(BCap<1>::SLAB == SLAB_BYTES, "the 4096-vertex instance keeps its slab layout");

typedef double d2 __attribute__((ext_vector_type(2)));

#ifdef MDQ_SB_TRACE
// debug build only: s_memtime cycles of the sections of a level (wave 0 of mesh 0), summed over the launch
static __device__ long long mdq_sb_trace_buf[8];
#define SB_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_sb_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
#else
#define SB_STAMP(k)
#endif

template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double grp8_sum(double v) {
  v += dppd<0xB1>(v);
  v += dppd<0x4E>(v);
  v += dppd<0x141>(v);
  return v;
}
__device__ __forceinline__ double grp8_min(double v) {
  v = fmin(v, dppd<0xB1>(v));
  v = fmin(v, dppd<0x4E>(v));
  v = fmin(v, dppd<0x141>(v));
  return v;
}
__device__ __forceinline__ double rsqrt_d(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ double rcp_d(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// the same update by a group of FOUR lanes from a work record whose entries are already in registers: lane l plays the lanes
// l and l + 4 of the 8-lane form (entries l, l + 8 and l + 4, l + 12 of the vertex's ascending-cell list), and the two partial
// sums go through the 8-lane tree - (xor 1, xor 2) inside each virtual quad, then quad 0 + quad 1 - so that every sum is
// associated exactly as in exact_vertex: identical bits, half the lanes (the level loop is bound by instruction issue: the
// scalar tail - divisions, comparisons - ran in eight lanes per vertex)
__device__ __forceinline__ d2 exact_vertex_rec(const d2* X, const double* r2k_tab, int v, int k, int l, const uint32_t (&w)[4]) {
#pragma clang fp contract(off)
  const double EPS = 3.0e-16;
  // every LDS read of the update is issued up front (the addresses are in registers: one round trip, not a chain of them);
  // entries beyond k read position 0 and contribute +0.0 / no minimum (x + 0.0 == x: the sums keep their bits)
  const d2 p = X[v];
  const double r2k = r2k_tab[k];                             // 1.0 / (2.0 * k), rounded once (the table is built with that division)
  // (entries l + 8, l + 12 exist for vertices with more than 8 cells only: a wave none of whose vertices has them skips
  //  that half of the arithmetic - the level loop is bound by the fp64 instructions it issues)
  const bool wide = __builtin_amdgcn_ballot_w64(k > 8) != 0;
  d2 pa[4], pc[4];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    pa[e] = X[w[e] & 0xFFFF];
    pc[e] = X[w[e] >> 16];
  }
  if (wide) {
#pragma unroll
    for (int e = 2; e < 4; ++e) {
      pa[e] = X[w[e] & 0xFFFF];
      pc[e] = X[w[e] >> 16];
    }
  }
  double sx[2] = {0.0, 0.0}, sy[2] = {0.0, 0.0}, rm = 1e300;
  auto entry = [&](int h, int e, bool on) {
    sx[h] += on ? pa[e].x + pc[e].x : 0.0;
    sy[h] += on ? pa[e].y + pc[e].y : 0.0;
    const double tx = pc[e].x - pa[e].x, ty = pc[e].y - pa[e].y;
    const double cr = ty * (p.x - pa[e].x) - tx * (p.y - pa[e].y);
    const double t2 = tx * tx + ty * ty;
    const double r_ = cr * cr * rcp_d(on ? t2 : 1.0);
    rm = on ? fmin(rm, r_) : rm;
  };
  entry(0, 0, l < k);                 // virtual lane l:     entry l
  if (wide) entry(0, 2, l + 8 < k);   //                     entry l + 8
  entry(1, 1, l + 4 < k);             // virtual lane l + 4: entry l + 4
  if (wide) entry(1, 3, l + 12 < k);  //                     entry l + 12
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sx[h] += dppd<0xB1>(sx[h]);
    sx[h] += dppd<0x4E>(sx[h]);
    sy[h] += dppd<0xB1>(sy[h]);
    sy[h] += dppd<0x4E>(sy[h]);
  }
  rm = fmin(rm, dppd<0xB1>(rm));
  rm = fmin(rm, dppd<0x4E>(rm));
  const double sxt = sx[0] + sx[1], syt = sy[0] + sy[1];
  const double dx = sxt * r2k - p.x, dy = syt * r2k - p.y;
  const double q2 = dx * dx + dy * dy;
  if (!(q2 >= EPS * EPS && q2 > 0.0)) return p;
  if (0.25 * rm < q2) {
    const double f = 0.5 * (rm * rsqrt_d(rm)) * rsqrt_d(q2);
    return d2{p.x + f * dx, p.y + f * dy};
  }
  return d2{p.x + dx, p.y + dy};
}

// DOLFIN's update of vertex v by a group of 8 lanes (lane l of the group takes incident cells l, l + 8, ...): the same
// arithmetic, in the same order, as exact_vertex of mdq_smooth_linear.hip / exact_update of mdq_smooth.hip
__device__ __forceinline__ d2 exact_vertex(const d2* X, const int* ptr, const uint32_t* inc, int v, int l) {
#pragma clang fp contract(off)
  const double EPS = 3.0e-16;
  const int q0 = ptr[v], k = ptr[v + 1] - q0;
  const d2 p = X[v];
  double sx = 0.0, sy = 0.0, rm = 1e300;
  for (int q = l; q < k; q += 8) {
    const uint32_t w = inc[q0 + q];
    const d2 pa = X[w & 0xFFFF], pc = X[w >> 16];
    sx += pa.x + pc.x;
    sy += pa.y + pc.y;
    const double tx = pc.x - pa.x, ty = pc.y - pa.y;
    const double cr = ty * (p.x - pa.x) - tx * (p.y - pa.y);
    rm = fmin(rm, cr * cr * rcp_d(tx * tx + ty * ty));   // SQUARED distance to the line through the opposite edge
  }
  sx = grp8_sum(sx);
  sy = grp8_sum(sy);
  rm = grp8_min(rm);
  const double r2k = 1.0 / (2.0 * k);
  const double dx = sx * r2k - p.x, dy = sy * r2k - p.y;
  const double q2 = dx * dx + dy * dy;
  if (!(q2 >= EPS * EPS && q2 > 0.0)) return p;          // |c - p| < DOLFIN_EPS: the vertex stays
  if (0.25 * rm < q2) {                                  // limited step: needs the lengths
    const double f = 0.5 * (rm * rsqrt_d(rm)) * rsqrt_d(q2);
    return d2{p.x + f * dx, p.y + f * dy};
  }
  return d2{p.x + dx, p.y + dy};                         // |c - p| <= r_min / 2: to the centroid itself
}

// The sweeps of the level schedule (shared by the two kernels below): positions X and level starts `lstart` in LDS, the
// level-ordered work records meta[i] = v | k << 16, inc2[i][MAXK] in the workspace.
template <bool GX = false>       // GX: the positions are in GLOBAL memory (the 16 384-vertex instance): a full barrier per level
__device__ __forceinline__ void level_sweeps(d2* X, const double* r2k_tab, const int* lstart, const uint32_t* meta, const uint32_t* inc2,
                                             int lmax, int S, int tid) {
  {
    // records of the first pass of the NEXT level are requested before the current level is computed: their latency hides
    // behind the update + the barrier (they do not depend on the positions)
    constexpr int G = BWG / 4;
    const int g4 = tid >> 2, l4 = tid & 3;
    struct Rec { uint32_t m, w[4]; };
    auto fetch = [&](int i, int i1, Rec& r) {
      r.m = 0u;
      r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
      if (i < i1) {
        r.m = meta[i];
        const uint32_t* e = inc2 + i * MAXK + l4;
        r.w[0] = e[0];      // entry l
        r.w[1] = e[4];      // entry l + 4
        r.w[2] = e[8];      // entry l + 8
        r.w[3] = e[12];     // entry l + 12
      }
    };
    // two record sets used alternately (a register copy `cur = nxt` would wait for the loads it copies - in front of the
    // barrier, i.e. the L2 round trip back on every level)
    Rec ra, rb;
    int i0 = lstart[1], i1 = lstart[2];                       // this level
    int j0 = lmax > 1 ? lstart[2] : i0, j1 = lmax > 1 ? lstart[3] : i1;   // the next one (level 1 again behind the last)
    fetch(i0 + g4, i1, ra);
    auto level = [&](int l, Rec& mine, Rec& next) {
      // the level starts two levels ahead are read now and used by the NEXT call's record request: no LDS round trip in
      // front of the global one
      const int l2 = (l % lmax + 1) % lmax + 1;              // the level after the next one (cyclic)
#ifdef MDQ_SB_TRACE
      long long tq_ = __builtin_amdgcn_s_memtime();
#endif
      const int k0 = lstart[l2], k1 = lstart[l2 + 1];
#ifndef MDQ_SB_NOFETCH
      fetch(j0 + g4, j1, next);
#endif
      SB_STAMP(0)
      for (int i = i0 + g4; i < i1; i += G) {
        if (i != i0 + g4) fetch(i, i1, mine);                 // (further passes of a wide level)
        const int v = mine.m & 0xFFFF, k = mine.m >> 16;
#ifdef MDQ_SB_NOCOMP
        const d2 xn = X[v] + X[mine.w[0] & 0xFFFF] * 1e-300;
        (void)k;
#else
        const d2 xn = exact_vertex_rec(X, r2k_tab, v, k, l4, mine.w);
#endif
        if (l4 == 0) X[v] = xn;
      }
      SB_STAMP(1)
      i0 = j0; i1 = j1; j0 = k0; j1 = k1;
      SB_STAMP(2)
      // LDS-only barrier: the positions of this level are in LDS (lgkmcnt), the records requested for the next level are
      // still on their way (vmcnt) and stay in flight - __syncthreads() drains the vector-memory counter as well
#ifndef MDQ_SB_NOBAR
      if (GX) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      SB_STAMP(3)
    };
    const int total = S * lmax;
    int t = 0, l = 1;
    for (; t + 1 < total; t += 2) {
      level(l, ra, rb);
      l = l < lmax ? l + 1 : 1;
      level(l, rb, ra);
      l = l < lmax ? l + 1 : 1;
    }
    if (t < total) level(l, ra, rb);
  }
}

// The sweeps of a SKEWED schedule (smooth_flow_kernel): every interior vertex has a time tau(v) >= 1 with
//      1 <= tau(w) - tau(v) <= P - 1      for every pair of interior neighbours v < w,
// and the update of v in sweep s runs at time step s P + tau(v) - 1.  The left inequality is the order of one Gauss-Seidel
// sweep (w needs the new position of its lower-numbered neighbour v); the right one lets sweep s + 1 START while sweep s is
// still running: v may take its sweep-(s + 1) update only after w has read v's sweep-s position, P + tau(v) > tau(w).  Two
// vertices that share a time step are never neighbours (their taus are equal or differ by a multiple of P), so a step updates
// in place like a level does - same values, same bits as the sequential sweeps.  tau = the level of the one-sweep DAG and
// P = the number of levels is the plain level schedule (50 x 40 = 2 000 steps on the red-refined ys930); with P = 20
// (smooth_flow_kernel finds the taus by relaxation) the same 50 sweeps are 1 020 steps twice as wide - the 968 levels of the
// DAG of all sweeps that tools/sweep_dag.py counted, without a work queue.  Records are ordered by (residue (tau - 1) mod P,
// then (tau - 1) / P): what a time step t updates - residue t mod P, the classes j whose sweep t / P - j is in [0, S) - is ONE
// contiguous range; `lstart[l]`, l = residue * J + j + 1, are the class starts.
__device__ __forceinline__ void skew_sweeps(d2* X, const double* r2k_tab, const int* lstart, const uint32_t* meta, const uint32_t* inc2,
                                            int P, int J, int S, int tid) {
  constexpr int G = BWG / 4;
  const int g4 = tid >> 2, l4 = tid & 3;
  struct Rec { uint32_t m, w[4]; };
  auto fetch = [&](int i, int i1, Rec& r) {
    r.m = 0u;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
    if (i < i1) {
      r.m = meta[i];
      const uint32_t* e = inc2 + i * MAXK + l4;
      r.w[0] = e[0];
      r.w[1] = e[4];
      r.w[2] = e[8];
      r.w[3] = e[12];
    }
  };
  const int T = (S - 1 + J) * P;                             // time steps
  // the record range of time step (q P + r)
  auto range = [&](int r, int q, int& a, int& b) {
    const int jlo = max(0, q - S + 1), jhi = min(J - 1, q);
    a = b = 0;
    if (jlo <= jhi && q < S - 1 + J) {
      a = lstart[r * J + jlo + 1];
      b = lstart[r * J + jhi + 2];
    }
  };
  Rec ra, rb;
  int i0, i1, j0, j1;
  range(0, 0, i0, i1);
  int r2 = P > 1 ? 1 : 0, q2 = P > 1 ? 0 : 1;                // (residue, sweep block) of the step after the current one ...
  range(r2, q2, j0, j1);
  fetch(i0 + g4, i1, ra);
  auto step = [&](Rec& mine, Rec& next) {
    // ... and of the one after that: its range is read now and used by the NEXT call's record request
    if (++r2 == P) { r2 = 0; ++q2; }
    int k0, k1;
    range(r2, q2, k0, k1);
    fetch(j0 + g4, j1, next);
    for (int i = i0 + g4; i < i1; i += G) {
      if (i != i0 + g4) fetch(i, i1, mine);                 // (further passes of a wide step)
      const int v = mine.m & 0xFFFF, k = mine.m >> 16;
      const d2 xn = exact_vertex_rec(X, r2k_tab, v, k, l4, mine.w);
      if (l4 == 0) X[v] = xn;
    }
    i0 = j0; i1 = j1; j0 = k0; j1 = k1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (LDS-only barrier: see level_sweeps)
  };
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(ra, rb);
    step(rb, ra);
  }
  if (t < T) step(ra, rb);
}

template <int KV>
__global__ __launch_bounds__(BWG) void smooth_big_kernel(int NV, int NT, double* coords, const int32_t* cells, const int32_t* nv_,
                                                         const int32_t* nt_, const int32_t* iters_, const int32_t* rem,
                                                         const int32_t* rstat, int iters_env, unsigned char* slab, int32_t* status) {
  constexpr int BNV = BCap<KV>::NV, BNT = BCap<KV>::NT;       // (shadow the 4096-vertex constants of the namespace)
  constexpr size_t SLAB_BYTES = BCap<KV>::SLAB;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_[];
  unsigned char* xbase = KV == 1 ? dyn_ : slab + (size_t)blockIdx.x * ((SLAB_BYTES + 255) & ~(size_t)255) + ((BCap<KV>::TABLES + 15) & ~(size_t)15);
  d2* X = reinterpret_cast<d2*>(xbase);                      // [BNV] 64 KB of LDS (KV = 1) / on the slab behind the tables
  int* cnt = reinterpret_cast<int*>(xbase + sizeof(d2) * BNV);  // [BNV] 16 KB: counts / cursors, then levels
  __shared__ int part[BWG];
  __shared__ int misc[4];
  __shared__ double r2k_tab[MAXK + 1];
  __shared__ int lstart[MAXLEV + 2];                          // level starts (a copy of lptr: the sweeps never leave the CU for them)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int S = iters_ ? iters_[b] : ((rem[b] >= 0 && rstat[b] == 0) ? iters_env : 0);
  if (status && tid == 0) status[b] = 0;
  if (S <= 0) return;
  const int nv = nv_[b], nt = nt_[b];
  double2* xg = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  const int32_t* tri = cells + (int64_t)b * NT * 3;
  unsigned char* sb = slab + (size_t)b * ((SLAB_BYTES + 255) & ~(size_t)255);
  int* ptr = reinterpret_cast<int*>(sb);                                   // [BNV + 1] vertex -> cells
  uint32_t* inc = reinterpret_cast<uint32_t*>(ptr + BNV + 8);              // [3 BNT] a | c << 16, sorted by cell per vertex
  uint32_t* cel = inc + 3 * BNT;                                           // [3 BNT] cell id of the entry
  uint16_t* order = reinterpret_cast<uint16_t*>(cel + 3 * BNT);            // [BNV] interior vertices grouped by level
  uint16_t* intr = order + BNV;                                            // [BNV] 1 = interior
  int* lptr = reinterpret_cast<int*>(intr + BNV);                          // [levels + 2]
  uint32_t* meta = reinterpret_cast<uint32_t*>(lptr + BNV + 8);            // [BNV] per level-ordered position: v | k << 16
  uint32_t* inc2 = meta + BNV;                                             // [BNV][MAXK] its entries, ascending cell
  // ---- vertex -> cells
  for (int v = tid; v < BNV; v += BWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += BWG)
    for (int k = 0; k < 3; ++k) atomicAdd(&cnt[tri[3 * t + k]], 1);
  __syncthreads();
  {   // exclusive scan of cnt[0 .. BNV) -> ptr: 8 entries per thread, thread totals by Hillis-Steele
    constexpr int PER = BNV / BWG;
    int loc[PER], run = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      loc[i] = run;
      run += cnt[tid * PER + i];
    }
    part[tid] = run;
    __syncthreads();
    for (int off = 1; off < BWG; off <<= 1) {
      const int add = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    const int base = part[tid] - run;
#pragma unroll
    for (int i = 0; i < PER; ++i) ptr[tid * PER + i] = base + loc[i];
    if (tid == BWG - 1) ptr[BNV] = part[tid];
  }
  __syncthreads();
  for (int v = tid; v < BNV; v += BWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += BWG) {
    const int vs[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k], a = vs[(k + 1) % 3], c = vs[(k + 2) % 3];
      const int q = ptr[v] + atomicAdd(&cnt[v], 1);
      inc[q] = (uint32_t)a | ((uint32_t)c << 16);
      cel[q] = (uint32_t)t;
    }
  }
  __syncthreads();
  // ---- per vertex: entries in ascending cell order; interior = every neighbour seen exactly twice
  for (int v = tid; v < nv; v += BWG) {
    const int q0 = ptr[v], k = ptr[v + 1] - q0;
    for (int i = 1; i < k; ++i) {
      const uint32_t ci = cel[q0 + i], wi = inc[q0 + i];
      int j = i - 1;
      while (j >= 0 && cel[q0 + j] > ci) {
        cel[q0 + j + 1] = cel[q0 + j];
        inc[q0 + j + 1] = inc[q0 + j];
        --j;
      }
      cel[q0 + j + 1] = ci;
      inc[q0 + j + 1] = wi;
    }
    bool interior = k > 0;
    for (int e = 0; e < 2 * k && interior; ++e) {
      const uint32_t we = inc[q0 + (e >> 1)];
      const uint32_t id = (e & 1) ? we >> 16 : we & 0xFFFF;
      int seen = 0;
      for (int f = 0; f < 2 * k; ++f) {
        const uint32_t wf = inc[q0 + (f >> 1)];
        seen += ((f & 1) ? wf >> 16 : wf & 0xFFFF) == id;
      }
      interior = seen == 2;
    }
    intr[v] = interior ? 1 : 0;
  }
  for (int v = tid; v < BNV; v += BWG) cnt[v] = 0;           // levels (0: fixed vertex / not reached yet)
  if (tid == 0) misc[0] = 1;
  __syncthreads();
  // ---- levels by relaxation: level(v) = 1 + max level of the lower-numbered interior neighbours (monotone: converges to the
  // longest-path levels in at most as many rounds as there are levels)
  int rounds = 0;
  while (misc[0] && rounds < BNV) {
    __syncthreads();
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    bool changed = false;
    for (int v = tid; v < nv; v += BWG) {
      if (!intr[v]) continue;
      const int q0 = ptr[v], k = ptr[v + 1] - q0;
      int lv = 1;
      for (int q = 0; q < k; ++q) {
        const uint32_t w = inc[q0 + q];
        const int a = w & 0xFFFF, c = w >> 16;
        if (a < v && intr[a]) lv = max(lv, cnt[a] + 1);
        if (c < v && intr[c]) lv = max(lv, cnt[c] + 1);
      }
      if (lv != cnt[v]) changed = true;
      cnt[v] = lv;                                           // (benign race: levels only grow towards the fixed point)
    }
    if (changed) misc[0] = 1;
    ++rounds;
    __syncthreads();
  }
  // ---- vertices grouped by level (counting sort; the order inside a level does not matter)
  int lmax = 0;
  for (int v = tid; v < nv; v += BWG) lmax = max(lmax, cnt[v]);
  part[tid] = lmax;
  __syncthreads();
  for (int off = BWG / 2; off > 0; off >>= 1) {
    if (tid < off) part[tid] = max(part[tid], part[tid + off]);
    __syncthreads();
  }
  lmax = part[0];
  __syncthreads();
  for (int l = tid; l <= lmax + 1; l += BWG) lptr[l] = 0;
  __syncthreads();
  for (int v = tid; v < nv; v += BWG)
    if (cnt[v] > 0) atomicAdd(&lptr[cnt[v] + 1], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int l = 0; l <= lmax + 1; ++l) {
      run += lptr[l];
      lptr[l] = run;
    }
  }
  __syncthreads();
  // (fill: cursors in `part` would not hold a level each - use the level start from lptr through atomics on a copy in cel,
  //  which is dead now)
  int* cur = reinterpret_cast<int*>(cel);
  for (int l = tid; l <= lmax; l += BWG) cur[l] = lptr[l];
  __syncthreads();
  for (int v = tid; v < nv; v += BWG)
    if (cnt[v] > 0) order[atomicAdd(&cur[cnt[v]], 1)] = (uint16_t)v;
  if (tid == 0) misc[1] = lmax > MAXLEV ? 1 : 0;             // 1: the general loop below (records / level starts do not fit)
  __syncthreads();
  // ---- work records in level order: the sweeps read them with addresses that depend on the position only (the walk
  // order[i] -> ptr[v] -> inc[q] was three dependent L2 round trips per update: 2 us per level, 4 ms per launch)
  const int n_int = lptr[lmax + 1];
  for (int i = tid; i < n_int; i += BWG) {
    const int v = order[i], q0 = ptr[v], k = ptr[v + 1] - q0;
    meta[i] = (uint32_t)v | ((uint32_t)k << 16);
    if (k > MAXK) misc[1] = 1;
    for (int q = 0; q < MAXK; ++q) inc2[i * MAXK + q] = q < k ? inc[q0 + q] : 0u;
  }
  for (int l = tid; l <= lmax + 1 && l < MAXLEV + 2; l += BWG) lstart[l] = lptr[l];
  if (tid >= 1 && tid <= MAXK) {
#pragma clang fp contract(off)
    r2k_tab[tid] = 1.0 / (2.0 * tid);
  }
  // ---- positions
  for (int v = tid; v < nv; v += BWG) {
    const double2 xv = xg[v];
    X[v] = d2{xv.x, xv.y};
  }
  __syncthreads();
  // ---- sweeps: level by level, 8 lanes per vertex
  const int grp = tid >> 3, l8 = tid & 7;
  const int nthr = BWG;
  if (lmax == 0) {
    // (no interior vertex: nothing moves)
  } else if (!misc[1]) {
    level_sweeps<(KV > 1)>(X, r2k_tab, lstart, meta, inc2, lmax, S, tid);
  } else
  for (int s = 0; s < S; ++s) {
    for (int l = 1; l <= lmax; ++l) {
      const int i0 = lptr[l], i1 = lptr[l + 1];
      for (int i = i0 + grp; i < i1; i += BWG / 8) {
        const int v = order[i];
        const d2 xn = exact_vertex(X, ptr, inc, v, l8);
        if (l8 == 0) X[v] = xn;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int v = tid; v < nv; v += nthr) {
    const d2 p = X[v];
    xg[v] = double2{p.x, p.y};
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same level schedule with the SET-UP OUT OF LDS (round 5), for meshes whose tables fit the LDS next to the positions
// (23 NV + 12 NT bytes <= ~151 KB: the red-refined ys930 of BASELINE configs[4] needs 148 KB): vertex -> cells CSR, per-vertex
// cell order, interior test, levels by relaxation and the counting sort by level all run on LDS arrays (in the kernel above
// they are loops of L2 round trips: 0.62 ms of its 2.2 ms; here 0.37 ms, of which the 41 relaxation rounds are 0.2 ms), the
// level-ordered work records go to the workspace and the sweeps are the shared level_sweeps().  A mesh with more than MAXK
// cells at a vertex, more than 1 000 levels or 255+ sweeps is handed back to the kernel above (`redo`).
//
// Measured and NOT kept (tools/sweep_dag.py: the dependency DAG of ALL 50 sweeps is 968 levels deep on this mesh against
// 50 x 40 = 2 000 - sweep s + 1 may start long before sweep s has ended - so two forms without a level schedule were built on
// this set-up; each group of four lanes owns the tasks (sweep, position in level order) g, g + G, ... in that order, a byte per
// vertex in LDS counts its finished sweeps, a task runs when its neighbours' counters allow it; same bits as the level form):
//   free-running waves, every wave polling (all neighbour bytes per round / one byte per round):   3.80 / 4.26 ms per launch
//   rounds with a workgroup barrier, every group running at most one ready task per round:         5.37 ms
// against 1.96 ms for the level schedule.  Free-running, the sixteen groups of a wave become ready a few hundred cycles apart
// and each one costs the wave a whole pass through the update code; in rounds, a group's list is in (sweep, level) order and
// its level-1 tasks of sweep s + 1 sit behind its last task of sweep s - the overlap of sweeps the DAG allows needs tasks
// picked out of order (a work queue), which was not built.
constexpr int FWG = 1024;
__host__ __device__ inline size_t flow_lds_bytes(int NV, int NT) {
  const size_t nvp = (size_t)((NV + 3) & ~3);
  return 16 * nvp + 12 * (size_t)NT + 2 * (nvp + 4) + 4 * nvp + nvp + 64;      // X | inc | ptr | trec | done
}

__global__ __launch_bounds__(FWG) void smooth_flow_kernel(int NV, int NT, double* coords, const int32_t* cells, const int32_t* nv_,
                                                          const int32_t* nt_, const int32_t* iters_, const int32_t* rem,
                                                          const int32_t* rstat, int iters_env, unsigned char* slab, int32_t* redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_[];
  const int nvp = (NV + 3) & ~3;
  d2* X = reinterpret_cast<d2*>(dyn_);                                       // [nvp]   (set-up: cnt int[NV] | cel u16[3 NT])
  uint32_t* inc = reinterpret_cast<uint32_t*>(dyn_ + 16 * (size_t)nvp);      // [3 NT]  a | c << 16, ascending cell per vertex
  uint16_t* ptr = reinterpret_cast<uint16_t*>(inc + 3 * (size_t)NT);          // [nvp + 4]
  uint32_t* trec = reinterpret_cast<uint32_t*>(ptr + nvp + 4);               // [nvp]   by interior rank: v | k << 12 | q0 << 17
  unsigned char* done = reinterpret_cast<unsigned char*>(trec + nvp);        // [nvp]   finished sweeps (255: fixed vertex)
  int* cnt = reinterpret_cast<int*>(dyn_);
  uint16_t* cel = reinterpret_cast<uint16_t*>(dyn_ + 4 * (size_t)nvp);
  __shared__ int part[FWG];
  __shared__ int lvl[MAXLEV + 2];                            // level counts / starts / cursors
  __shared__ double r2k_tab[MAXK + 1];
  __shared__ int misc[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int S = iters_ ? iters_[b] : ((rem[b] >= 0 && rstat[b] == 0) ? iters_env : 0);
  // redo[b]: sweeps handed back to the level kernel (launched behind this one; it returns at once for 0) - a mesh with more
  // than MAXK cells at a vertex or a sweep count that does not fit the byte counters
  if (tid == 0) redo[b] = (S >= 255) ? S : 0;
  if (S <= 0 || S >= 255) return;
  const int nv = nv_[b], nt = nt_[b];
  double2* xg = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  const int32_t* tri = cells + (int64_t)b * NT * 3;
  // block-wide exclusive scan of a[0 .. nvp) (ints in LDS), four consecutive entries per thread; returns the total
  auto scan_excl = [&](int* a) {
    const int per = (nvp + FWG - 1) / FWG;        // <= 4 for NV <= 4096
    int loc[4], run = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid * per + i;
      loc[i] = run;
      run += (i < per && idx < nvp) ? a[idx] : 0;
    }
    part[tid] = run;
    __syncthreads();
    for (int off = 1; off < FWG; off <<= 1) {
      const int add = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    const int base = part[tid] - run, total = part[FWG - 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid * per + i;
      if (i < per && idx < nvp) a[idx] = base + loc[i];
    }
    __syncthreads();
    return total;
  };
  // ---- vertex -> cells
  for (int v = tid; v < nvp; v += FWG) cnt[v] = 0;
  if (tid == 0) misc[0] = 0;
  if (tid >= 1 && tid <= MAXK) {
#pragma clang fp contract(off)
    r2k_tab[tid] = 1.0 / (2.0 * tid);
  }
  __syncthreads();
  for (int t = tid; t < nt; t += FWG)
    for (int k = 0; k < 3; ++k) atomicAdd(&cnt[tri[3 * t + k]], 1);
  __syncthreads();
  scan_excl(cnt);
  for (int v = tid; v < nvp; v += FWG) ptr[v] = (uint16_t)cnt[v];
  if (tid == 0) ptr[nvp] = (uint16_t)(3 * nt);
  __syncthreads();
  // (the cursors: cnt[v] itself runs from ptr[v] upwards)
  for (int t = tid; t < nt; t += FWG) {
    const int vs[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k], a = vs[(k + 1) % 3], c = vs[(k + 2) % 3];
      const int q = atomicAdd(&cnt[v], 1);
      inc[q] = (uint32_t)a | ((uint32_t)c << 16);
      cel[q] = (uint16_t)t;
    }
  }
  __syncthreads();
  // ---- per vertex: entries in ascending cell order; interior = every neighbour seen exactly twice
  for (int v = tid; v < nvp; v += FWG) {
    unsigned char dn = 255;
    if (v < nv) {
      const int q0 = ptr[v], k = (v + 1 < nvp ? (int)ptr[v + 1] : 3 * nt) - q0;
      for (int i = 1; i < k; ++i) {
        const uint16_t ci = cel[q0 + i];
        const uint32_t wi = inc[q0 + i];
        int j = i - 1;
        while (j >= 0 && cel[q0 + j] > ci) {
          cel[q0 + j + 1] = cel[q0 + j];
          inc[q0 + j + 1] = inc[q0 + j];
          --j;
        }
        cel[q0 + j + 1] = ci;
        inc[q0 + j + 1] = wi;
      }
      bool interior = k > 0;
      for (int e = 0; e < 2 * k && interior; ++e) {
        const uint32_t we = inc[q0 + (e >> 1)];
        const uint32_t id = (e & 1) ? we >> 16 : we & 0xFFFF;
        int seen = 0;
        for (int f = 0; f < 2 * k; ++f) {
          const uint32_t wf = inc[q0 + (f >> 1)];
          seen += ((f & 1) ? wf >> 16 : wf & 0xFFFF) == id;
        }
        interior = seen == 2;
      }
      if (interior) dn = 0;
      if (interior && k > MAXK) misc[0] = 1;             // (more than 16 cells at a vertex: the level kernel's general loop)
    }
    done[v] = dn;
  }
  __syncthreads();
  // ---- levels of one sweep's dependency DAG by relaxation (level(v) = 1 + max over the lower-numbered interior neighbours;
  // monotone, converges in as many rounds as there are levels), then the task records in LEVEL order: (sweep, level, position)
  // is a topological order of the DAG of all sweeps, and dealing the positions out round-robin puts the vertices of a level
  // on different groups (in index order a group's next vertex was rarely the one the wavefront had reached: 4.0 ms)
  int* lev = cnt;                                            // (cel is dead: the scratch may be rewritten)
  // (Kahn's algorithm instead of relaxation - every vertex is expanded ONCE, when its level is known: the 41 relaxation rounds,
  //  each re-reading every vertex's entries through three dependent LDS reads, were 0.2 ms of the set-up)
  int* indeg = reinterpret_cast<int*>(cel);                  // [nvp] lower interior neighbours still without a level (x 2: every
  for (int v = tid; v < nvp; v += FWG) {                     //  neighbour appears in two cells)
    int n_low = 0;
    if (v < nv && done[v] == 0) {
      const int q0 = ptr[v], k = (v + 1 < nvp ? (int)ptr[v + 1] : 3 * nt) - q0;
      for (int q = 0; q < k; ++q) {
        const uint32_t w_ = inc[q0 + q];
        const int a_ = w_ & 0xFFFF, c_ = w_ >> 16;
        n_low += (a_ < v && done[a_] == 0) + (c_ < v && done[c_] == 0);
      }
    }
    lev[v] = 0;
    indeg[v] = n_low;
  }
  __syncthreads();
  for (int level = 1; level <= nvp; ++level) {
    int found = 0;
    for (int v = tid; v < nv; v += FWG)
      if (done[v] == 0 && lev[v] == 0 && indeg[v] == 0) {
        lev[v] = level;
        found = 1;
      }
    if (!__syncthreads_or(found)) break;
    for (int v = tid; v < nv; v += FWG)
      if (lev[v] == level && done[v] == 0) {
        const int q0 = ptr[v], k = (v + 1 < nvp ? (int)ptr[v + 1] : 3 * nt) - q0;
        for (int q = 0; q < k; ++q) {
          const uint32_t w_ = inc[q0 + q];
          const int a_ = w_ & 0xFFFF, c_ = w_ >> 16;
          if (a_ > v && done[a_] == 0) atomicSub(&indeg[a_], 1);
          if (c_ > v && done[c_] == 0) atomicSub(&indeg[c_], 1);
        }
      }
    __syncthreads();
  }
  __syncthreads();
  // ---- number of levels; then the SKEW (skew_sweeps): times tau(v) with 1 <= tau(w) - tau(v) <= P - 1 over the interior edges
  // v < w for a period P below the number of levels, by raise-only relaxation from the levels (the levels are the least
  // solution of the left inequalities; raising tau(v) to tau(w) - (P - 1) where the right one fails and re-establishing the
  // left ones converges to the least solution above them when one exists - in place: a monotone iteration from below reaches
  // the same fixed point in any order).  P = half the levels (rounded up), then 5/8 and 3/4 of them; no P: tau = level, P =
  // levels (the plain level schedule).  MDQ_NO_SMOOTH_SKEW (compile time): never.
  int lmax = 0;
  for (int v = tid; v < nv; v += FWG) lmax = max(lmax, done[v] == 0 ? lev[v] : 0);
  part[tid] = lmax;
  __syncthreads();
  for (int off = FWG / 2; off > 0; off >>= 1) {
    if (tid < off) part[tid] = max(part[tid], part[tid + off]);
    __syncthreads();
  }
  lmax = part[0];
  __syncthreads();
  int P = lmax > 0 ? lmax : 1, J = 1;
#ifndef MDQ_NO_SMOOTH_SKEW
  if (lmax >= 8 && lmax <= 1000) {
    int* tau = indeg;                                        // (the in-degrees are all zero now)
    const int cand[3] = {(lmax + 1) / 2, (5 * lmax + 7) / 8, (3 * lmax + 3) / 4};
    for (int c = 0; c < 3 && J == 1 && P == lmax; ++c) {
      const int Pt = cand[c], lim = 3 * lmax;                // (a tau beyond three times the levels: no solution for this P)
      for (int v = tid; v < nvp; v += FWG) tau[v] = lev[v];
      __syncthreads();
      bool ok = false;
      for (int round = 0; round < 6 * lmax; ++round) {
        int changed = 0;
        for (int v = tid; v < nv; v += FWG)
          if (done[v] == 0) {
            const int q0 = ptr[v], k = (v + 1 < nvp ? (int)ptr[v + 1] : 3 * nt) - q0;
            int t0 = tau[v], t_ = t0;
            for (int q = 0; q < k; ++q) {
              const uint32_t w_ = inc[q0 + q];
              const int a_ = w_ & 0xFFFF, c_ = w_ >> 16;
              if (done[a_] == 0) t_ = max(t_, a_ < v ? tau[a_] + 1 : tau[a_] - (Pt - 1));
              if (done[c_] == 0) t_ = max(t_, c_ < v ? tau[c_] + 1 : tau[c_] - (Pt - 1));
            }
            if (t_ != t0) {
              tau[v] = min(t_, lim + 1);
              changed = max(changed, t_ > lim ? 2 : 1);
            }
          }
        const int any = __syncthreads_or(changed);
        if (!any) {
          ok = true;
          break;
        }
        if ((round & 7) == 7 && __syncthreads_or(changed == 2)) break;   // diverging (looked at every eighth round: tau is capped)
      }
      if (ok) {
        int tmax = 0;
        for (int v = tid; v < nv; v += FWG) tmax = max(tmax, done[v] == 0 ? tau[v] : 0);
        part[tid] = tmax;
        __syncthreads();
        for (int off = FWG / 2; off > 0; off >>= 1) {
          if (tid < off) part[tid] = max(part[tid], part[tid + off]);
          __syncthreads();
        }
        tmax = part[0];
        __syncthreads();
        const int Jt = (tmax + Pt - 1) / Pt;
        if (Pt * Jt <= 1000) {
          P = Pt;
          J = Jt;
          // the class of a vertex takes the place of its level: residue-major, 1-based
          for (int v = tid; v < nv; v += FWG)
            if (done[v] == 0) lev[v] = ((tau[v] - 1) % P) * J + (tau[v] - 1) / P + 1;
        }
      }
      __syncthreads();
    }
  }
#endif
  for (int l_ = tid; l_ < MAXLEV + 2; l_ += FWG) lvl[l_] = 0;
  __syncthreads();
  for (int v = tid; v < nv; v += FWG)
    if (done[v] == 0) {
      if (lev[v] > 1000) misc[0] = 1;                      // (level starts live in `part`: FWG ints)
      else atomicAdd(&lvl[lev[v]], 1);
    }
  __syncthreads();
  if (tid == 0) {                                            // exclusive scan over the classes / levels in use (a few dozen)
    int run = 0;
    const int nl_ = min(MAXLEV + 2, P * J + 2);
    for (int l_ = 0; l_ < nl_; ++l_) {
      const int c_ = lvl[l_];
      lvl[l_] = run;
      run += c_;
    }
    misc[2] = run;
  }
  __syncthreads();
  const int n_int = misc[2];
  if (misc[0] == 0)
    for (int v = tid; v < nv; v += FWG)
      if (done[v] == 0) {
        const int q0 = ptr[v], k = (v + 1 < nvp ? (int)ptr[v + 1] : 3 * nt) - q0;
        trec[atomicAdd(&lvl[lev[v]], 1)] = (uint32_t)v | ((uint32_t)k << 12) | ((uint32_t)q0 << 17);
      }
  __syncthreads();
  if (misc[0] != 0) {                   // (handed back: the coordinates are untouched)
    if (tid == 0) redo[b] = S;
    return;
  }
  // ---- the level-ordered work records of the sweeps, to the workspace (the sweeps read them one level ahead); level starts
  uint32_t* meta = reinterpret_cast<uint32_t*>(slab + (size_t)b * ((SLAB_BYTES + 255) & ~(size_t)255));
  uint32_t* inc2 = meta + BNV;
  for (int i = tid; i < n_int; i += FWG) {
    const uint32_t tr = trec[i];
    const int v = tr & 0xFFF, k = (tr >> 12) & 31, q0 = tr >> 17;
    meta[i] = (uint32_t)v | ((uint32_t)k << 16);
    for (int q = 0; q < MAXK; ++q) inc2[i * MAXK + q] = q < k ? inc[q0 + q] : 0u;
  }
  // (after the fill lvl[l] is the END of class / level l = the start of l + 1; 0 is empty)
  const int ncls = P * J;                                    // classes (= levels when P is their number)
  __syncthreads();
  for (int l_ = tid; l_ <= ncls + 1; l_ += FWG) part[l_] = l_ >= 1 ? lvl[l_ - 1] : 0;      // lstart (part: MAXLEV + 2 <= FWG ints)
  __syncthreads();
  // ---- positions (over the set-up scratch: the barriers above end its last readers)
  for (int v = tid; v < nv; v += FWG) {
    const double2 xv = xg[v];
    X[v] = d2{xv.x, xv.y};
  }
  __syncthreads();
  if (lmax > 0) skew_sweeps(X, r2k_tab, part, meta, inc2, P, J, S, tid);
  __syncthreads();
  for (int v = tid; v < nv; v += FWG) {
    const d2 p = X[v];
    xg[v] = double2{p.x, p.y};
  }
}
}  // namespace mdq_smooth_big

#ifdef MDQ_SB_TRACE
extern "C" MDQ_API int mdq_sb_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_smooth_big::mdq_sb_trace_buf), sizeof(long long) * 8) != hipSuccess) return -1;
  if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_smooth_big::mdq_sb_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif

// bytes of caller workspace the large-mesh kernel needs (0: the mesh fits the 1024-vertex kernels; -1: beyond the kernels)
static int64_t smooth_big_workspace_bytes(int32_t B, int32_t NV, int32_t NT) {
  using namespace mdq_smooth_big;
  if (NV > BCap<4>::NV || NT > BCap<4>::NT) return -1;
  if (NV > BNV || NT > BNT) return (int64_t)((BCap<4>::SLAB + 255) & ~(size_t)255) * B;                // (no hand-back launch there)
  return (int64_t)((SLAB_BYTES + 255) & ~(size_t)255) * B + (((int64_t)B * 4 + 255) & ~(int64_t)255);   // tables + the hand-back counts
}

// mdq_smooth / mdq_smooth_fast / mdq_smooth_fast_env for NV > 1024 (called by those entry points)
static int smooth_big_launch(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                             const int32_t* nt, const int32_t* iterations, const int32_t* rem, const int32_t* rstat,
                             int32_t iters_env, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace mdq_smooth_big;
  if (NV > BCap<4>::NV || NT > BCap<4>::NT) return mdq_set_error("mdq_smooth: capacity above 16384 vertices / 32768 triangles");
  if (!workspace || workspace_bytes < smooth_big_workspace_bytes(B, NV, NT) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return mdq_set_error("mdq_smooth: workspace missing, too small or not 16-byte aligned (mdq_smooth_workspace_bytes / mdq_smooth_fast_workspace_bytes)");
  unsigned char* slab = static_cast<unsigned char*>(workspace);
  if (NV > BNV || NT > BNT) {      // the 16 384-vertex instance: positions on the slab, no LDS beyond the static tables
    hipLaunchKernelGGL(smooth_big_kernel<4>, dim3(B), dim3(BWG), 0, (hipStream_t)stream, NV, NT, coords, cells, nv, nt, iterations, rem,
                       rstat, iters_env, slab, nullptr);
    if (hipGetLastError() != hipSuccess) return mdq_set_error("smooth_big_kernel<4> launch failed");
    return 0;
  }
  const size_t lds = sizeof(d2) * BNV + sizeof(int) * BNV;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&smooth_big_kernel<1>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(d2) * BNV + sizeof(int) * BNV));
  if (attr != hipSuccess) return mdq_set_error("hipFuncSetAttribute(smooth_big_kernel) failed");
  // meshes whose tables fit the LDS beside the positions: the kernel with the LDS set-up, then this launch for what it hands back
  // (normally nothing: every workgroup returns at once).  MDQ_NO_SMOOTH_FLOW=1: the level kernel alone (A / B switch)
  const size_t flow_lds = flow_lds_bytes(NV, NT);
  static const bool no_flow = std::getenv("MDQ_NO_SMOOTH_FLOW") != nullptr;
  if (!no_flow && flow_lds + 8832 <= 160 * 1024 && NT <= 2 * ((NV + 3) & ~3)) {
    static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&smooth_flow_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8832);
    if (attr2 != hipSuccess) return mdq_set_error("hipFuncSetAttribute(smooth_flow_kernel) failed");
    int32_t* redo = reinterpret_cast<int32_t*>(slab + ((SLAB_BYTES + 255) & ~(size_t)255) * (size_t)B);
    hipLaunchKernelGGL(smooth_flow_kernel, dim3(B), dim3(FWG), flow_lds, (hipStream_t)stream, NV, NT, coords, cells, nv, nt,
                       iterations, rem, rstat, iters_env, slab, redo);
    hipLaunchKernelGGL(smooth_big_kernel<1>, dim3(B), dim3(BWG), lds, (hipStream_t)stream, NV, NT, coords, cells, nv, nt, redo,
                       nullptr, nullptr, 0, slab, nullptr);
    if (hipGetLastError() != hipSuccess) return mdq_set_error("smooth_flow_kernel launch failed");
    return 0;
  }
  hipLaunchKernelGGL(smooth_big_kernel<1>, dim3(B), dim3(BWG), lds, (hipStream_t)stream, NV, NT, coords, cells, nv, nt, iterations,
                     rem, rstat, iters_env, slab, nullptr);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("smooth_big_kernel launch failed");
  return 0;
}
