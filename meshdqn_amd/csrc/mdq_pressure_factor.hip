// Device-side build of the substructuring factors of the pressure matrix (the factorisation half of the reference's
// `LUSolver("mumps")`, flow_solver.py:150-159, which it repeats for every coarsened mesh, :318-328): what
// meshdqn_amd/pressure_direct.py does in numpy on the host, as ONE 1024-thread workgroup per environment -
//   recursive coordinate bisection into 8 balanced parts (ranks by counting), vertex separator (the higher-part end of
//   every coupling between parts), node ordering (interiors by part, then the separator; ballot scans),
//   per subdomain: dense K_II in LDS -> in-place Gauss-Jordan inverse W_s (SPD: no pivoting), F_s = W_s K_IG,
//   Schur complement S -= K_GI F_s (global, one subdomain after the other), at the end S -> LDS -> inverse.
// The outputs are the pd_* arrays of mdq_ipcs_desc in exactly the layout the solve phase (pressure_direct in
// mdq_ipcs.hip) reads; the matrix is the scaled, boundary-eliminated P1 stiffness in SELL-64 as written by
// mdq_ipcs_assemble / mdq_ipcs_setup_matfree.  An environment whose mesh exceeds the LDS-sized limits below gets
// nparts = 0 in its header (the pressure kernel then runs its CG) and a negative status.
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"

namespace mdq_pf {

#ifdef MDQ_PF_TRACE
__device__ long long mdq_pf_trace_buf[16];
#define PF_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_pf_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
extern "C" int mdq_pf_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_pf_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_pf_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define PF_STAMP(k)
#endif

constexpr int TH = 1024;      // threads (16 waves)
constexpr int PARTS = 8;      // subdomains (3 bisection levels)
constexpr int MMAX = 112;     // interior nodes of a subdomain (K_II: 112^2 doubles = 98 KB of LDS)
constexpr int GMAX = 48;      // separator nodes touching one subdomain (K_IG: 112 x 48 doubles = 42 KB)
constexpr int NGMAX = 112;    // separator nodes (S takes the K_II buffer for its inversion)
constexpr int NVMAX = 1024;

struct Scan {
  int* wtot;   // [TH / 64 + 1]
};

// exclusive scan of a 0/1 flag over the workgroup; total in `total` (two barriers)
__device__ __forceinline__ int scan_flag(bool flag, int* wtot, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wtot[wave] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < TH / 64; ++w) {
    const int t = wtot[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + pre;
}

__device__ __forceinline__ unsigned long long ord_key(double x) {   // order-preserving map double -> uint64
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// in-place inverse of the SPD m x m matrix A (column-major, LDS, m <= MMAX) by Gauss-Jordan with the matrix held in
// REGISTERS for the m elimination steps: thread (i, jg) = (tid & 127, tid >> 7) owns row i of the columns jg, jg + 8, ...
// (14 doubles).  A step is one rank-1 update A -= c (x) r: the owners of row k / column k publish them through LDS
// (double-buffered by the parity of k: ONE barrier per step), and the special cases of the textbook formulas
//    a_kj' = a_kj / p,  a_ik' = -a_ik / p,  a_kk' = 1 / p,  a_ij' = a_ij - a_ik a_kj / p
// come out of the same update with c_k := p - 1 and r_k := p + 1 in place of p (then c_i r_j / p gives all four).
// rc: 2 x (128 + 128 + 2) doubles of scratch.
constexpr int NQ = (MMAX + 7) / 8;
__device__ inline void invert_spd(double* A, int m, double* rc) {
  const int tid = threadIdx.x, i = tid & 127, jg = tid >> 7;
  double a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = jg + 8 * q;
    a[q] = (i < m && j < m) ? A[j * m + i] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int kq = 0; kq < NQ; ++kq) {
    for (int kj = 0; kj < 8; ++kj) {
      const int k = kq * 8 + kj;
      if (k >= m) break;
      double* rowb = rc + (k & 1) * 258;     // row k (raw; entry k holds p + 1)
      double* colb = rowb + 128;             // column k (raw; entry k holds p - 1)
      if (i == k) {   // (entries past m are zeros: the update below needs no bounds)
#pragma unroll
        for (int q = 0; q < NQ; ++q) rowb[jg + 8 * q] = (q == kq && jg == kj) ? a[q] + 1.0 : a[q];
      }
      if (jg == kj) {
        colb[i] = i == k ? a[kq] - 1.0 : a[kq];
        if (i == k) colb[128] = 1.0 / a[kq];
      }
      __syncthreads();
      {
        // branch-free: all 14 row entries are fetched together (rows / columns past m hold zeros and stay zero)
        const double t = colb[i] * colb[128];   // c_i / p
        double r[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) r[q] = rowb[jg + 8 * q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) a[q] = fma(-t, r[q], a[q]);
        if (i == k && jg == kj) a[kq] = colb[128];   // (the pivot's 1 / p without the cancellation of p - (p^2 - 1) / p)
      }
    }
  }
  __syncthreads();
  // back to LDS, then symmetrise (what the host does after numpy's inverse)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = jg + 8 * q;
    if (i < m && j < m) A[j * m + i] = a[q];
  }
  __syncthreads();
  if (i < m)
    for (int j = jg; j < m; j += TH / 128)
      if (i < j) {
        const double sv = 0.5 * (A[j * m + i] + A[i * m + j]);
        A[j * m + i] = sv;
        A[i * m + j] = sv;
      }
  __syncthreads();
}

__global__ __launch_bounds__(TH) void pressure_factor_kernel(mdq_ipcs_desc d, int32_t* status) {
  extern __shared__ __align__(16) double sm[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t B = b;
  const int nv = d.nv[b];
  // outputs of this environment
  int32_t* hdr = const_cast<int32_t*>(d.pd_hdr) + B * 4;
  int32_t* o_node = const_cast<int32_t*>(d.pd_node) + B * d.NV;
  int32_t* o_meta = const_cast<int32_t*>(d.pd_meta) + B * d.NPART * 6;
  int32_t* o_rowblk = const_cast<int32_t*>(d.pd_rowblk) + B * d.NV;
  double* o_W = const_cast<double*>(d.pd_W) + B * d.NPW;
  double* o_F = const_cast<double*>(d.pd_F) + B * d.NPF;
  int32_t* o_gidx = const_cast<int32_t*>(d.pd_gidx) + B * d.NPGI;
  double* o_S = const_cast<double*>(d.pd_Sinv) + B * d.NPS;
  int32_t* o_gkp = const_cast<int32_t*>(d.pd_gk_ptr) + B * (d.NV + 1);
  int32_t* o_gkc = const_cast<int32_t*>(d.pd_gk_col) + B * d.NPGK;
  double* o_gkv = const_cast<double*>(d.pd_gk_val) + B * d.NPGK;
  // the matrix
  const int32_t* so = d.sl1_off + B * (d.NV / 64 + 2);
  const int32_t* sc = d.sl1_col + B * d.NSE1;
  const double* K = d.K1s + B * d.NSE1;
  const double* xg = d.coords + B * d.NV * 2;
  // LDS
  double* BIG = sm;                               // [MMAX * MMAX]: keys of the bisection, then K_II / F_s, then S
  double* KIG = BIG + MMAX * MMAX;                // [MMAX * GMAX]
  double* rk = KIG + MMAX * GMAX;                 // [2 x 258] row / column / 1 / p of an elimination step, by parity
  unsigned long long* ext = reinterpret_cast<unsigned long long*>(rk + 520);   // [PARTS][4] min x, max x, min y, max y
  int* gsize = reinterpret_cast<int*>(ext + PARTS * 4);                        // [PARTS]
  int* wtot = gsize + PARTS;                                                   // [TH / 64]
  int* misc = wtot + TH / 64;                                                  // [16]
  uint16_t* inv = reinterpret_cast<uint16_t*>(misc + 16);                      // [NVMAX] permuted position of a node
  uint16_t* node = inv + NVMAX;                                                // [NVMAX]
  uint16_t* gpos = node + NVMAX;                                               // [NGMAX + 16] column of a separator node in K_IG
  uint8_t* part = reinterpret_cast<uint8_t*>(gpos + NGMAX + 16);               // [NVMAX]
  uint8_t* sepf = part + NVMAX;                                                // [NVMAX]
  uint8_t* gflag = sepf + NVMAX;                                               // [NGMAX + 16]
  uint8_t* glist = gflag + NGMAX + 16;                                         // [GMAX + 16] separator-local id of K_IG column k

  auto fail = [&](int code) {
    if (tid == 0) {
      hdr[0] = hdr[1] = hdr[2] = hdr[3] = 0;
      if (status) status[b] = code;
    }
  };
  if (nv > NVMAX || nv < 1) {
    fail(-1);
    return;
  }
  // real entries of SELL row r: columns ascend, the padding repeats the row index
  auto for_row = [&](int r, auto&& fn) {
    const int off = so[r >> 6], width = (so[(r >> 6) + 1] - off) >> 6;
    int prev = -1;
    for (int j = 0; j < width; ++j) {
      const int ps = off + j * 64 + (r & 63);
      const int c = sc[ps];
      if (c > prev) {
        prev = c;
        const double val = K[ps];
        if (val != 0.0) fn(c, val);
      }
    }
  };

#ifdef MDQ_PF_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  // ---------------------------------------------------------------- recursive coordinate bisection
  double* cx = BIG;
  double* cy = BIG + NVMAX;
  double* key = BIG + 2 * NVMAX;
  for (int v = tid; v < nv; v += TH) {
    cx[v] = xg[2 * v];
    cy[v] = xg[2 * v + 1];
    part[v] = 0;
    sepf[v] = 0;
  }
  __syncthreads();
  for (int level = 0; level < 3; ++level) {
    const int ng = 1 << level;
    if (tid < PARTS * 4) ext[tid] = (tid & 1) ? 0ull : ~0ull;   // max slots start at 0, min slots at all ones
    if (tid < PARTS) gsize[tid] = 0;
    __syncthreads();
    for (int v = tid; v < nv; v += TH) {
      const int g = part[v];
      const unsigned long long kx = ord_key(cx[v]), ky = ord_key(cy[v]);
      atomicMin(&ext[g * 4 + 0], kx);
      atomicMax(&ext[g * 4 + 1], kx);
      atomicMin(&ext[g * 4 + 2], ky);
      atomicMax(&ext[g * 4 + 3], ky);
      atomicAdd(&gsize[g], 1);
    }
    __syncthreads();
    for (int v = tid; v < nv; v += TH) {
      const int g = part[v];
      // extents from the ordered keys back to doubles
      auto back = [](unsigned long long u) {
        const unsigned long long w = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
        return __longlong_as_double((long long)w);
      };
      const double ex = back(ext[g * 4 + 1]) - back(ext[g * 4 + 0]), ey = back(ext[g * 4 + 3]) - back(ext[g * 4 + 2]);
      key[v] = ex >= ey ? cx[v] : cy[v];       // (argmax of the extents: x on a tie)
    }
    __syncthreads();
    int np_ = 0;
    if (tid < nv) {
      const int v = tid, g = part[v];
      const double kv = key[v];
      int r = 0;
      for (int u = 0; u < nv; ++u) {
        const double ku = key[u];
        r += (part[u] == g) & ((ku < kv) | ((ku == kv) & (u < v)));
      }
      const int sz = gsize[g];
      np_ = 2 * g + ((sz > 1 && r >= sz / 2) ? 1 : 0);
    }
    __syncthreads();
    if (tid < nv) part[tid] = (uint8_t)np_;
    (void)ng;
    __syncthreads();
  }
  PF_STAMP(0)
  // ---------------------------------------------------------------- vertex separator
  for (int v = tid; v < nv; v += TH) {
    bool s = false;
    const int pv = part[v];
    for_row(v, [&](int c, double) { s = s || (c != v && part[c] < pv); });
    sepf[v] = s ? 1 : 0;
  }
  __syncthreads();
  // ---------------------------------------------------------------- ordering: interiors by part, then the separator
  int nI = 0, nparts = 0;
  int q0s[PARTS], ms[PARTS];
  for (int s = 0; s < PARTS; ++s) {
    const bool f = tid < nv && part[tid] == s && !sepf[tid];
    int tot;
    const int pos = scan_flag(f, wtot, tot);
    if (tot > 0) {
      if (f) {
        node[nI + pos] = (uint16_t)tid;
        inv[tid] = (uint16_t)(nI + pos);
        o_node[nI + pos] = tid;
        o_rowblk[nI + pos] = nparts;
      }
      q0s[nparts] = nI;
      ms[nparts] = tot;
      ++nparts;
      nI += tot;
    }
  }
  int nG;
  {
    const bool f = tid < nv && sepf[tid];
    const int pos = scan_flag(f, wtot, nG);
    if (f) {
      node[nI + pos] = (uint16_t)tid;
      inv[tid] = (uint16_t)(nI + pos);
      o_node[nI + pos] = tid;
    }
  }
  __syncthreads();
  {
    int mmax = 0;
    for (int s = 0; s < nparts; ++s) mmax = max(mmax, ms[s]);
    if (mmax > MMAX || nG > NGMAX || nparts > d.NPART || (int64_t)nG * nG > d.NPS) {
      fail(-2);
      return;
    }
  }
  PF_STAMP(1)
  // ---------------------------------------------------------------- K[G, I] as CSR (permuted interior numbering), S = K[G, G]
  for (int idx = tid; idx < nG * nG; idx += TH) o_S[idx] = 0.0;
  __threadfence();
  {
    int cnt = 0;
    const int v = tid < nG ? node[nI + tid] : 0;
    if (tid < nG) for_row(v, [&](int c, double) { cnt += sepf[c] ? 0 : 1; });
    // exclusive scan of the counts over the (at most NGMAX <= 128) separator rows: two waves, serial over wave totals
    int* cs = reinterpret_cast<int*>(key);   // (the bisection keys are dead)
    if (tid < 128) cs[tid] = tid < nG ? cnt : 0;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int g = 0; g < nG; ++g) {
        const int c = cs[g];
        cs[g] = run;
        run += c;
      }
      cs[nG] = run;
    }
    __syncthreads();
    if (cs[nG] > d.NPGK) {
      fail(-3);
      return;
    }
    if (tid <= nG) o_gkp[tid] = cs[tid];
    if (tid < nG) {
      int p = cs[tid];
      for_row(v, [&](int c, double val) {
        if (sepf[c]) {
          o_S[(int64_t)(inv[c] - nI) * nG + tid] = val;     // column-major (row tid, column of c)
        } else {
          o_gkc[p] = inv[c];
          o_gkv[p] = val;
          ++p;
        }
      });
    }
  }
  __threadfence();
  __syncthreads();
  PF_STAMP(2)
  // ---------------------------------------------------------------- subdomains
  int woff = 0, foff = 0, gioff = 0;
  for (int s = 0; s < nparts; ++s) {
    const int q0 = q0s[s], m = ms[s];
    for (int idx = tid; idx < m * m; idx += TH) BIG[idx] = 0.0;
    for (int g = tid; g < nG; g += TH) gflag[g] = 0;
    __syncthreads();
    for (int li = tid; li < m; li += TH) {
      const int v = node[q0 + li];
      for_row(v, [&](int c, double val) {
        if (sepf[c])
          gflag[inv[c] - nI] = 1;
        else
          BIG[(inv[c] - q0) * m + li] = val;     // (an interior only couples to its own subdomain and the separator)
      });
    }
    __syncthreads();
    int gs;
    {
      const bool f = tid < nG && gflag[tid];
      const int pos = scan_flag(f, wtot, gs);
      if (f) gpos[tid] = (uint16_t)pos;
      if (gs <= GMAX && gioff + gs <= d.NPGI && f) {
        o_gidx[gioff + pos] = tid;
        glist[pos] = (uint8_t)tid;
      }
    }
    if (gs > GMAX || woff + m * m > d.NPW || foff + m * gs > d.NPF || gioff + gs > d.NPGI) {
      fail(-4);
      return;
    }
    for (int idx = tid; idx < m * gs; idx += TH) KIG[idx] = 0.0;
    __syncthreads();
    for (int li = tid; li < m; li += TH) {
      const int v = node[q0 + li];
      for_row(v, [&](int c, double val) {
        if (sepf[c]) KIG[gpos[inv[c] - nI] * m + li] = val;
      });
    }
    __syncthreads();
    PF_STAMP(3)
    invert_spd(BIG, m, rk);
    PF_STAMP(4)
    for (int idx = tid; idx < m * m; idx += TH) o_W[woff + idx] = BIG[idx];
    // F_s = W_s K_IG (registers first: F takes the K_II buffer afterwards)
    constexpr int FPT = (MMAX * GMAX + TH - 1) / TH;
    double fv[FPT];
#pragma unroll
    for (int q = 0; q < FPT; ++q) {
      const int idx = tid + q * TH;
      double acc = 0.0;
      if (idx < m * gs) {
        const int k = idx / m, i = idx - k * m;
        for (int j = 0; j < m; ++j) acc += BIG[j * m + i] * KIG[k * m + j];
      }
      fv[q] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FPT; ++q) {
      const int idx = tid + q * TH;
      if (idx < m * gs) {
        BIG[idx] = fv[q];
        o_F[foff + idx] = fv[q];
      }
    }
    __syncthreads();
    PF_STAMP(5)
    // S[loc, loc] -= K_GI F_s  (device-scope atomics: the element may have been written by another thread earlier)
    for (int idx = tid; idx < gs * gs; idx += TH) {
      const int bq = idx / gs, a = idx - bq * gs;
      double acc = 0.0;
      for (int i = 0; i < m; ++i) acc += KIG[a * m + i] * BIG[bq * m + i];
      atomicAdd(&o_S[(int64_t)glist[bq] * nG + glist[a]], -acc);     // separator-local ids of the two columns
    }
    if (tid == 0) {
      int32_t* m6 = o_meta + 6 * s;
      m6[0] = q0; m6[1] = m; m6[2] = woff; m6[3] = foff; m6[4] = gs; m6[5] = gioff;
    }
    woff += m * m;
    foff += m * gs;
    gioff += gs;
    __threadfence();
    __syncthreads();
    PF_STAMP(6)
  }
  // ---------------------------------------------------------------- inverse Schur complement
  for (int idx = tid; idx < nG * nG; idx += TH)
    BIG[idx] = __hip_atomic_load(&o_S[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (nG > 0) invert_spd(BIG, nG, rk);
  for (int idx = tid; idx < nG * nG; idx += TH) o_S[idx] = BIG[idx];
  PF_STAMP(7)
  if (tid == 0) {
    hdr[0] = nI;
    hdr[1] = nG;
    hdr[2] = nparts;
    hdr[3] = 0;
    if (status) status[b] = 0;
  }
}

}  // namespace mdq_pf

extern "C" int mdq_ipcs_factorize_pressure(const mdq_ipcs_desc* d, int32_t* status, void* stream) {
  using namespace mdq_pf;
  if (!d || d->B <= 0 || !d->pd_hdr || !d->pd_node || !d->pd_meta || !d->pd_rowblk || !d->pd_W || !d->pd_F || !d->pd_gidx ||
      !d->pd_Sinv || !d->pd_gk_ptr || !d->pd_gk_col || !d->pd_gk_val || !d->K1s || !d->sl1_off || !d->sl1_col || !d->coords)
    return mdq_set_error("mdq_ipcs_factorize_pressure: bad arguments");
  if (d->NPART < 1) return mdq_set_error("mdq_ipcs_factorize_pressure: NPART < 1");
  const size_t lds = sizeof(double) * ((size_t)MMAX * MMAX + (size_t)MMAX * GMAX + 520) + sizeof(unsigned long long) * PARTS * 4 +
                     sizeof(int) * (PARTS + TH / 64 + 16) + sizeof(uint16_t) * (2 * NVMAX + NGMAX + 16) + 2 * NVMAX + NGMAX + 16 + GMAX + 16 + 64;
  if (lds > 160 * 1024) return mdq_set_error("mdq_ipcs_factorize_pressure: LDS plan exceeds 160 KB");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pressure_factor_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  hipLaunchKernelGGL(pressure_factor_kernel, dim3(d->B), dim3(TH), lds, (hipStream_t)stream, *d, status);
  e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  return 0;
}
