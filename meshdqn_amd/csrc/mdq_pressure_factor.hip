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
#include "mdq_internal.h"

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

// in-place inverse of the SPD m x m matrix A (column-major, LDS, m <= MMAX = 112) by Gauss-Jordan with the matrix held in
// REGISTERS for the m elimination steps: thread (ti, tj) = (tid % 28, tid / 28) of the first 784 owns the 4 x 4 tile of
// rows 4 ti .. and columns 4 tj ..; a step is one rank-1 update A -= c (x) r / p for which a thread fetches 4 + 4 values
// (four 16-byte LDS reads for 16 FMAs: with one row per thread it was 15 reads for 14 FMAs, and the 16 waves of the
// workgroup were bound by the LDS bandwidth of their one CU).  The owners of row k / column k publish them through LDS,
// double-buffered by the parity of k: ONE barrier per step.  The special cases of the textbook formulas
//    a_kj' = a_kj / p,  a_ik' = -a_ik / p,  a_kk' = 1 / p,  a_ij' = a_ij - a_ik a_kj / p
// come out of the same update with c_k := p - 1 and r_k := p + 1 in place of p.
// rc: 2 x (128 + 128 + 2) doubles of scratch (16-byte aligned).
constexpr int TL = MMAX / 4;   // 28 tiles per side
static_assert(MMAX % 4 == 0 && TL * TL <= TH, "tile layout");
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ inline void invert_spd(double* A, int m, double* rc) {
  const int tid = threadIdx.x;
  const bool act = tid < TL * TL;
  const int ti = act ? tid % TL : 0, tj = act ? tid / TL : 0;
  double a[4][4];   // a[r][c] = A(4 ti + r, 4 tj + c)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = 4 * ti + r, j = 4 * tj + c;
      a[r][c] = (act && i < m && j < m) ? A[j * m + i] : 0.0;
    }
  __syncthreads();
  const int mt = (m + 3) >> 2;
  for (int kt = 0; kt < mt; ++kt) {
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      const int k = 4 * kt + kr;
      if (k >= m) break;
      double* rowb = rc + (k & 1) * 258;     // row k (raw; entry k holds p + 1; zeros past m)
      double* colb = rowb + 128;             // column k (raw; entry k holds p - 1; zeros past m); [128] = 1 / p
      if (act && ti == kt) {                 // this tile holds row k: its four entries of it
#pragma unroll
        for (int c = 0; c < 4; ++c) rowb[4 * tj + c] = (tj == kt && c == kr) ? a[kr][c] + 1.0 : a[kr][c];
      }
      if (act && tj == kt) {                 // ... column k
#pragma unroll
        for (int r = 0; r < 4; ++r) colb[4 * ti + r] = (ti == kt && r == kr) ? a[r][kr] - 1.0 : a[r][kr];
        if (ti == kt) colb[128] = 1.0 / a[kr][kr];
      }
      __syncthreads();
      {
        const double ip = colb[128];
        const d2v c01 = *reinterpret_cast<const d2v*>(colb + 4 * ti), c23 = *reinterpret_cast<const d2v*>(colb + 4 * ti + 2);
        const d2v r01 = *reinterpret_cast<const d2v*>(rowb + 4 * tj), r23 = *reinterpret_cast<const d2v*>(rowb + 4 * tj + 2);
        const double cv[4] = {c01.x * ip, c01.y * ip, c23.x * ip, c23.y * ip};
        const double rv[4] = {r01.x, r01.y, r23.x, r23.y};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[r][c] = fma(-cv[r], rv[c], a[r][c]);
        if (act && ti == kt && tj == kt) a[kr][kr] = ip;   // (the pivot's 1 / p without the cancellation of p - (p^2 - 1) / p)
      }
    }
  }
  __syncthreads();
  // back to LDS, then symmetrise (what the host does after numpy's inverse)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = 4 * ti + r, j = 4 * tj + c;
      if (act && i < m && j < m) A[j * m + i] = a[r][c];
    }
  __syncthreads();
  {
    const int i = tid & 127, jg = tid >> 7;
    if (i < m)
      for (int j = jg; j < m; j += TH / 128)
        if (i < j) {
          const double sv = 0.5 * (A[j * m + i] + A[i * m + j]);
          A[j * m + i] = sv;
          A[i * m + j] = sv;
        }
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
  double* KIG = BIG + MMAX * MMAX;                // [(MMAX + 1) * GMAX]: K_IG, column stride ldk (odd: no LDS bank conflicts
                                                  // when the lanes of a wave read different columns)
  double* rk = KIG + (MMAX + 2) * GMAX;               // [2 x 258] row / column / 1 / p of an elimination step, by parity
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
    // composite sort key: part | top 51 bits of the order-preserving coordinate key | vertex id, so that ONE unsigned
    // comparison ranks a vertex inside its part (rank over all vertices minus the sizes of the smaller parts); a tie
    // of the truncated coordinates goes by the vertex id - the bisection stays exactly balanced either way
    unsigned long long* ckey = reinterpret_cast<unsigned long long*>(key);
    for (int v = tid; v < NVMAX; v += TH) {
      unsigned long long kk = ~0ull;           // (slots past nv: larger than every real key)
      if (v < nv) {
        const int g = part[v];
        auto back = [](unsigned long long u) {
          const unsigned long long w = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
          return __longlong_as_double((long long)w);
        };
        const double ex = back(ext[g * 4 + 1]) - back(ext[g * 4 + 0]), ey = back(ext[g * 4 + 3]) - back(ext[g * 4 + 2]);
        const unsigned long long ok = ord_key(ex >= ey ? cx[v] : cy[v]);       // (argmax of the extents: x on a tie)
        kk = ((unsigned long long)g << 61) | ((ok >> 13) << 10) | (unsigned long long)v;
      }
      ckey[v] = kk;
    }
    __syncthreads();
    int np_ = 0;
    if (tid < nv) {
      const int v = tid, g = part[v];
      const unsigned long long kv = ckey[v];
      int r = 0;
      const int nv4 = (nv + 3) & ~3;
      for (int u = 0; u < nv4; u += 4) {
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(ckey + u), k23 = *reinterpret_cast<const ulonglong2*>(ckey + u + 2);
        r += (k01.x < kv) + (k01.y < kv) + (k23.x < kv) + (k23.y < kv);
      }
      int below = 0;
      for (int q = 0; q < g; ++q) below += gsize[q];
      const int sz = gsize[g];
      np_ = 2 * g + ((sz > 1 && r - below >= sz / 2) ? 1 : 0);
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
    if (mmax > MMAX || nG > NGMAX || nparts > d.NPART || (int64_t)NGMAX * NGMAX + (int64_t)PARTS * GMAX * GMAX > d.NPS) {
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
  int gss[PARTS], gios[PARTS];
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
    const int ldk = m | 1;
    for (int idx = tid; idx < ldk * gs; idx += TH) KIG[idx] = 0.0;
    __syncthreads();
    for (int li = tid; li < m; li += TH) {
      const int v = node[q0 + li];
      for_row(v, [&](int c, double val) {
        if (sepf[c]) KIG[gpos[inv[c] - nI] * ldk + li] = val;
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
        for (int j = 0; j < m; ++j) acc += BIG[j * m + i] * KIG[k * ldk + j];
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
    // (the update is symmetric: the gs (gs + 1) / 2 pairs a <= bq, one per thread, written to both positions)
    for (int idx = tid; idx < gs * (gs + 1) / 2; idx += TH) {
      int bq = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (bq * (bq + 1) / 2 > idx) --bq;                   // (the float estimate may be off by one either way)
      while ((bq + 1) * (bq + 2) / 2 <= idx) ++bq;
      const int a = idx - bq * (bq + 1) / 2;
      const double* ka = KIG + a * ldk;
      const double* fb = BIG + bq * m;
      double acc0 = 0.0, acc1 = 0.0;
      int i = 0;
      for (; i + 1 < m; i += 2) {
        acc0 = fma(ka[i], fb[i], acc0);
        acc1 = fma(ka[i + 1], fb[i + 1], acc1);
      }
      if (i < m) acc0 = fma(ka[i], fb[i], acc0);
      const double acc = acc0 + acc1;
      // the subdomain's Schur term goes to a slice of its own (plain stores, nothing is read back here: a load would
      // wait behind the W / F stores of this subdomain, ~20 us); the slices are subtracted from K_GG at the end
      double* sl = o_S + NGMAX * NGMAX + s * GMAX * GMAX;
      sl[bq * gs + a] = acc;
      if (a != bq) sl[a * gs + bq] = acc;
    }
    if (tid == 0) {
      int32_t* m6 = o_meta + 6 * s;
      m6[0] = q0; m6[1] = m; m6[2] = woff; m6[3] = foff; m6[4] = gs; m6[5] = gioff;
    }
    gss[s] = gs;
    gios[s] = gioff;
    woff += m * m;
    foff += m * gs;
    gioff += gs;
    // (no fence here: nothing this subdomain wrote to global memory is read before the end of the loop; the barrier
    // only hands the LDS buffers to the next subdomain)
    __syncthreads();
    PF_STAMP(6)
  }
  __threadfence();      // the slices and separator lists written above are read back below
  __syncthreads();
  // ---------------------------------------------------------------- inverse Schur complement
  for (int idx = tid; idx < nG * nG; idx += TH)
    BIG[idx] = __hip_atomic_load(&o_S[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  for (int s = 0; s < nparts; ++s) {      // S = K_GG - sum of the subdomains' terms (in subdomain order: deterministic)
    const int gs = gss[s];
    const double* sl = o_S + NGMAX * NGMAX + s * GMAX * GMAX;
    const int32_t* gl = o_gidx + gios[s];
    for (int idx = tid; idx < gs * gs; idx += TH) {
      const int bq = idx / gs, a = idx - bq * gs;
      const int ga = __hip_atomic_load(&gl[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int gb = __hip_atomic_load(&gl[bq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      BIG[gb * nG + ga] -= __hip_atomic_load(&sl[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
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
  const size_t lds = sizeof(double) * ((size_t)MMAX * MMAX + (size_t)(MMAX + 2) * GMAX + 520) + sizeof(unsigned long long) * PARTS * 4 +
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
