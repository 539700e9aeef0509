// DOLFIN `Mesh.smooth(n)` (flow_solver.py:65-67, 236-237) on the GPU for B meshes at once.
//
// The algorithm is a Gauss-Seidel sweep over the interior vertices IN INDEX ORDER, repeated n times:
//   c = mean of the edge-connected neighbours, r_min = min distance from the vertex to the lines through the
//   opposite edges of its cells, p <- p + min(|c-p|, r_min/2) (c-p)/|c-p|.
// Update (sweep s, vertex v) needs neighbour w's value of sweep s if w < v and of sweep s-1 if w > v.  That
// dependency graph has only ~6 independent updates per level on the reference meshes (ys930: 34 700 updates in
// 5 454 levels), so the kernel is a latency-bound DATAFLOW machine, one 512-thread workgroup per mesh out of LDS:
// 64 groups of 8 lanes, group g owns the interior vertices of rank g, g+64, ... and walks them in
// (sweep, index) order; the 8 lanes take one incident cell each (sqrt + division per cell in parallel), combine
// with DPP, lane 0 publishes the new position and then the vertex's sweep counter; readiness is checked against
// the neighbours' counters.  The globally smallest unfinished update is always at the head of its group and
// always ready, so the machine cannot deadlock.  Cell contributions are summed in a fixed lane order: results
// are bitwise reproducible and agree with the sequential host loop to round-off (different association).
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"

namespace mdq_smoothing {

constexpr int SNV = 1024;      // vertex capacity (ids must fit 10 bits)
constexpr int SNT = 2048;      // triangle capacity (cell id must fit 12 bits)
#ifndef MDQ_SMOOTH_WG
#define MDQ_SMOOTH_WG 512
#endif
constexpr int SWG = MDQ_SMOOTH_WG;  // threads per workgroup: 8 waves measured best (256: 3.9 ms, 512: 3.4 ms, 1024: 3.9 ms for ys930)
                               // issue slots from the group on the critical path
#ifndef MDQ_SMOOTH_GRP
#define MDQ_SMOOTH_GRP 8
#endif
constexpr int GRP = MDQ_SMOOTH_GRP;  // lanes per vertex update (8 or 16)
constexpr int NGRP = SWG / GRP;
constexpr int PER = SNV / SWG; // entries per thread in the setup scans

template <int CTRL>
__device__ __forceinline__ double dpp8(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// all-reduce over groups of 8 consecutive lanes: xor 1, xor 2 (quad_perm), then the other quad (row_half_mirror)
__device__ __forceinline__ double grp_sum(double v) {
  v += dpp8<0xB1>(v);
  v += dpp8<0x4E>(v);
  v += dpp8<0x141>(v);
#if MDQ_SMOOTH_GRP == 16
  v += dpp8<0x140>(v);  // row_mirror: the other half of the row of 16
#endif
  return v;
}
// 1/sqrt(x) to double precision (not correctly rounded: ~1 ulp): hardware estimate + two Newton steps.  The IEEE
// sqrt + division sequences are ~25 dependent fp64 instructions each and every one of them is on the critical path
// of this latency-bound kernel.
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}

// 1/x to double precision (~1 ulp): hardware estimate + two Newton steps
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

__device__ __forceinline__ double grp_min(double v) {
  v = fmin(v, dpp8<0xB1>(v));
  v = fmin(v, dpp8<0x4E>(v));
  v = fmin(v, dpp8<0x141>(v));
#if MDQ_SMOOTH_GRP == 16
  v = fmin(v, dpp8<0x140>(v));
#endif
  return v;
}

// inclusive scan of data[0..SNV) in place (SWG threads, PER consecutive entries each; part = SWG ints of scratch)
__device__ __forceinline__ void scan_inclusive(int* data, int* part) {
  const int tid = threadIdx.x;
  int loc[PER], run = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    run += data[tid * PER + i];
    loc[i] = run;
  }
  part[tid] = run;
  __syncthreads();
  for (int off = 1; off < SWG; off <<= 1) {
    const int add = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  const int base = part[tid] - run;
#pragma unroll
  for (int i = 0; i < PER; ++i) data[tid * PER + i] = base + loc[i];
  __syncthreads();
}

__global__ __launch_bounds__(SWG) void smooth_kernel(int NV, int NT, double* coords, const int32_t* cells,
                                                     const int32_t* nv_, const int32_t* nt_, const int32_t* iters_,
                                                     long long* trace) {
#pragma clang fp contract(off)
  __shared__ double2 pos[SNV];
  __shared__ int done[SNV];
  __shared__ int ptr[SNV + 1];
  __shared__ int cnt[SNV];
  __shared__ uint32_t inc[3 * SNT];   // a | b << 10 | cell << 20 of every (vertex, incident cell), grouped by vertex, ascending cell
  __shared__ uint16_t ivert[SNV];     // interior vertices in index order
  __shared__ int part[SWG];
  __shared__ double rcp2k[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int iters = iters_[b];
  if (iters <= 0) return;
  const int nv = nv_[b], nt = nt_[b];
  double2* x = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  const int32_t* tri = cells + (int64_t)b * NT * 3;
  for (int v = tid; v < SNV; v += SWG) {
    if (v < nv) pos[v] = x[v];
    cnt[v] = 0;
  }
  if (tid < 32) rcp2k[tid] = tid ? 1.0 / (2.0 * tid) : 0.0;
  __syncthreads();
  for (int t = tid; t < nt; t += SWG)
    for (int k = 0; k < 3; ++k) atomicAdd(&cnt[tri[3 * t + k]], 1);
  __syncthreads();
  scan_inclusive(cnt, part);
  for (int v = tid; v < SNV; v += SWG) ptr[v + 1] = cnt[v];
  if (tid == 0) ptr[0] = 0;
  __syncthreads();
  for (int v = tid; v < SNV; v += SWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += SWG) {
    const int vs[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k], a = vs[(k + 1) % 3], c = vs[(k + 2) % 3];
      const int q = ptr[v] + atomicAdd(&cnt[v], 1);
      inc[q] = (uint32_t)a | ((uint32_t)c << 10) | ((uint32_t)t << 20);
    }
  }
  __syncthreads();
  // per vertex: order the incident cells by cell id (fixed summation order), boundary test (a neighbour seen once)
  for (int v = tid; v < SNV; v += SWG) {
    bool interior = false;
    if (v < nv) {
      const int q0 = ptr[v], q1 = ptr[v + 1];
      for (int i = q0 + 1; i < q1; ++i) {
        const uint32_t w = inc[i];
        int j = i - 1;
        while (j >= q0 && (inc[j] >> 20) > (w >> 20)) {
          inc[j + 1] = inc[j];
          --j;
        }
        inc[j + 1] = w;
      }
      interior = q1 > q0;
      for (int i = q0; i < q1 && interior; ++i)
        for (int h = 0; h < 2; ++h) {
          const uint32_t nb = h ? (inc[i] >> 10) & 0x3FF : inc[i] & 0x3FF;
          int seen = 0;
          for (int j = q0; j < q1; ++j) seen += ((inc[j] & 0x3FF) == nb) + (((inc[j] >> 10) & 0x3FF) == nb);
          if (seen != 2) interior = false;
        }
    }
    done[v] = interior ? 0 : 0x3FFFFFFF;
    cnt[v] = interior ? 1 : 0;
  }
  __syncthreads();
  scan_inclusive(cnt, part);
  const int n_int = cnt[SNV - 1];
  __syncthreads();
  // queue order = index order (ranking the vertices by their dependency level inside a sweep and dealing the levels
  // round-robin was tried: the setup costs more than the better-ordered queues gain)
  for (int v = tid; v < SNV; v += SWG)
    if (done[v] == 0) ivert[cnt[v] - 1] = (uint16_t)v;
  __syncthreads();

  // ---------------- dataflow Gauss-Seidel
  // group rank: consecutive ranks sit in DIFFERENT waves (they are usually neighbours on the dependency chain, and
  // the groups of one wave serialise whenever one of them computes)
  const int l = tid % GRP;
  const int g = (tid >> 6) + (SWG / 64) * ((tid & 63) / GRP);
  const int nown = g < n_int ? (n_int - g + NGRP - 1) / NGRP : 0;
  const double EPS = 3.0e-16;
  int s = 0, j = 0;
  const int total = nown * iters;
#define LD_DONE(i) __hip_atomic_load(&done[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LD_X(i) __hip_atomic_load(&pos[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LD_Y(i) __hip_atomic_load(&pos[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  // metadata of the update at the head of the group's queue (reloaded only when the head advances)
  int v = 0, k = 0, q0 = 0, a0 = 0, c0 = 0;
  bool fresh = true;
  for (int step = 0; step < total;) {
    if (fresh) {
      v = ivert[g + NGRP * j];
      q0 = ptr[v];
      k = ptr[v + 1] - q0;
      if (l < k) {
        const uint32_t w = inc[q0 + l];
        a0 = w & 0x3FF;
        c0 = (w >> 10) & 0x3FF;
      }
      fresh = false;
    }
    // poll: sweep counters of the neighbours in my cell(s) (cells l, l + 8, ...; the first one from registers)
    bool ok = true;
    if (l < k) {   // both counters loaded before either is tested: ONE LDS round trip per poll (a short-circuit `&&`
      const int da = LD_DONE(a0), dc = LD_DONE(c0);   // made the second load wait for the first, on the critical path)
      ok = (da >= (a0 < v ? s + 1 : s)) & (dc >= (c0 < v ? s + 1 : s));
    }
    for (int q = l + GRP; q < k; q += GRP) {
      const uint32_t w = inc[q0 + q];
      const int a = w & 0x3FF, c = (w >> 10) & 0x3FF;
      const int da = LD_DONE(a), dc = LD_DONE(c);
      ok = ok & (da >= (a < v ? s + 1 : s)) & (dc >= (c < v ? s + 1 : s));
    }
    const unsigned long long bal = __ballot(ok);
    constexpr unsigned long long GMASK = (1ull << GRP) - 1;
    const bool ready = ((bal >> ((tid & 63) & ~(GRP - 1))) & GMASK) == GMASK;
    if (ready) {
#ifdef MDQ_SMOOTH_TRACE
      const long long t_ready = clock64();
#endif
#ifndef MDQ_SMOOTH_NOPRIO
      // a wave that computes an update outranks the waves that only poll (they share the SIMD's issue slots)
      __builtin_amdgcn_s_setprio(3);
#endif
      // (a counter that is high enough guarantees that a position read after it is the right version: the
      // neighbour cannot advance again before this vertex has; LDS operations of a wave complete in order)
      const double px = LD_X(v), py = LD_Y(v);
      double sx = 0.0, sy = 0.0, rm = 1e300;
      for (int q = l; q < k; q += GRP) {
        int a = a0, c = c0;
        if (q != l) {
          const uint32_t w = inc[q0 + q];
          a = w & 0x3FF;
          c = (w >> 10) & 0x3FF;
        }
        const double ax = LD_X(a), ay = LD_Y(a), bx = LD_X(c), by = LD_Y(c);
        sx += ax + bx;
        sy += ay + by;
        const double tx = bx - ax, ty = by - ay;
#ifdef MDQ_SMOOTH_NOMATH
        rm = fmin(rm, fabs(ty * (px - ax) - tx * (py - ay)));
#else
        // SQUARED distance to the line through the opposite edge: the comparison with the step length below is done
        // on squares, so that the common case (full step to the centroid) needs no square root at all
        const double cr = ty * (px - ax) - tx * (py - ay);
        rm = fmin(rm, cr * cr * rcp_nr(tx * tx + ty * ty));
#endif
      }
      // lane 0 carries the x component, lane 1 the y component (same instruction stream: free)
      // the three group reductions step by step side by side (three independent dependency chains in flight instead
      // of one after the other; every lane takes part: DPP needs the whole group active)
      double tsx = sx, tsy = sy;
      {
        double ax_ = dpp8<0xB1>(tsx), ay_ = dpp8<0xB1>(tsy), am_ = dpp8<0xB1>(rm);
        tsx += ax_;
        tsy += ay_;
        rm = fmin(rm, am_);
        ax_ = dpp8<0x4E>(tsx);
        ay_ = dpp8<0x4E>(tsy);
        am_ = dpp8<0x4E>(rm);
        tsx += ax_;
        tsy += ay_;
        rm = fmin(rm, am_);
        ax_ = dpp8<0x141>(tsx);
        ay_ = dpp8<0x141>(tsy);
        am_ = dpp8<0x141>(rm);
        tsx += ax_;
        tsy += ay_;
        rm = fmin(rm, am_);
#if MDQ_SMOOTH_GRP == 16
        ax_ = dpp8<0x140>(tsx);
        ay_ = dpp8<0x140>(tsy);
        am_ = dpp8<0x140>(rm);
        tsx += ax_;
        tsy += ay_;
        rm = fmin(rm, am_);
#endif
      }
      const double sc = l == 0 ? tsx : tsy;
      const double pc = l == 0 ? px : py;
      const double dc_ = sc * rcp2k[k] - pc;                // lane 0: dx, lane 1: dy
      const double dother = dpp8<0xB1>(dc_);                // the other component
      const double q2 = dc_ * dc_ + dother * dother;        // (x*x + y*y in lane 0, y*y + x*x in lane 1: same bits)
      // rm holds the SQUARED minimum altitude.  |c - p| <= r_min / 2  <=>  q2 <= rm / 4: the vertex moves to the
      // centroid itself (p + d: the reference's p + r (d / r) up to one rounding); only a LIMITED step needs the
      // lengths (two reciprocal square roots, off the common path)
      if (l < 2) {
        if (q2 >= EPS * EPS && q2 > 0.0) {
          double pn = pc + dc_;
#ifndef MDQ_SMOOTH_NOMATH
          if (0.25 * rm < q2) {
            const double ir = rsqrt_nr(q2);
            const double rmin = rm * rsqrt_nr(rm);   // sqrt(rm)
            pn = pc + (0.5 * rmin * ir) * dc_;
          }
#endif
          __hip_atomic_store(l == 0 ? &pos[v].x : &pos[v].y, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      // (the position stores and the counter store below go to LDS from the SAME wave: the LDS executes a wave's
      // operations in issue order, so a compiler-level fence is enough - the workgroup-scope release fence waited
      // for the position stores to complete before the counter store could even be issued, on the critical path)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (l == 0) __hip_atomic_store(&done[v], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MDQ_SMOOTH_TRACE
      if (trace && b == 0 && l == 0) {
        trace[2 * ((int64_t)s * SNV + v)] = t_ready;
        trace[2 * ((int64_t)s * SNV + v) + 1] = clock64();
      }
#endif
      ++step;
      if (++j == nown) {
        j = 0;
        ++s;
      }
      fresh = true;
#ifndef MDQ_SMOOTH_NOPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
#ifdef MDQ_SMOOTH_SLEEP
    else if (bal == 0) {
      __builtin_amdgcn_s_sleep(MDQ_SMOOTH_SLEEP);
    }
#endif
  }
#undef LD_DONE
#undef LD_X
#undef LD_Y
  __syncthreads();
  for (int v = tid; v < nv; v += SWG) x[v] = pos[v];
}

}  // namespace mdq_smoothing

#ifdef MDQ_SMOOTH_TRACE
// debug builds only: [64 sweeps][SNV][2] shader-clock timestamps (ready, done) of every update of environment 0,
// in page-locked host memory the kernel writes directly
extern "C" long long* mdq_smooth_trace_host() {
  static long long* g_trace = nullptr;
  if (!g_trace) {
    hipHostMalloc(reinterpret_cast<void**>(&g_trace), sizeof(long long) * 2 * 64 * mdq_smoothing::SNV, hipHostMallocDefault);
    memset(g_trace, 0, sizeof(long long) * 2 * 64 * mdq_smoothing::SNV);
  }
  return g_trace;
}
#endif

extern "C" int mdq_smooth(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                          const int32_t* nt, const int32_t* iterations, void* stream) {
  if (B <= 0 || !coords || !cells || !nv || !nt || !iterations) return mdq_set_error("mdq_smooth: bad arguments");
  if (NV > mdq_smoothing::SNV || NT > mdq_smoothing::SNT)
    return mdq_set_error("mdq_smooth: mesh capacity above 1024 vertices / 2048 triangles (use mdq_smooth_host)");
  long long* trace = nullptr;
#ifdef MDQ_SMOOTH_TRACE
  trace = mdq_smooth_trace_host();
#endif
  hipLaunchKernelGGL(mdq_smoothing::smooth_kernel, dim3(B), dim3(mdq_smoothing::SWG), 0, (hipStream_t)stream, NV, NT, coords,
                     cells, nv, nt, iterations, trace);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("smooth_kernel launch failed");
  return 0;
}
