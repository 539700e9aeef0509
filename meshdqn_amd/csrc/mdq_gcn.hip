// Fused batched forward of the reference's graph Q-networks on gfx950.
//
// Kernel 1 (gcn_embed_kernel): ONE 256-thread workgroup per graph runs every conv / pool level out
// of LDS - CSR-by-target mean aggregation (SAGEConv), symmetric-normalised aggregation (GCNConv),
// TopKPooling (tanh score, rank-by-counting top-k, edge filter + relabel), [max || mean] readouts -
// and writes the 2C-wide graph embedding (sum of the per-level readouts).
// Kernel 2 (mlp_head_kernel): the dense head lin1 -> relu -> lin2 -> relu -> lin3 (-> softmax),
// batched over graphs on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// Restated semantics: airfoilgcnn.py:85-145 (NodeRemovalNet.forward), :170-209 (AirfoilGCNN.forward)
// and the PyG layers they call.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_gcn {

#ifndef MDQ_GCN_WG
#define MDQ_GCN_WG 512
#endif
constexpr int WGT = MDQ_GCN_WG;  // threads of the embedding kernel (one workgroup per graph)
constexpr int WGH = 256;         // threads of the MFMA head kernel (4 waves, one 32x32 block each per pass)
#ifndef MDQ_GCN_FORMC_MAXFIN
#define MDQ_GCN_FORMC_MAXFIN 32   // widest input the node-per-lane convolution (form (c)) takes
#endif
constexpr int NACC = 24;  // accumulators per thread of the "feature-outer" small-graph convolution

#ifdef MDQ_GCN_TRACE
// debug build only: s_memtime deltas of thread 0 of graph 0 at the phase boundaries, [level][phase]
__device__ long long mdq_gcn_trace_buf[8 * 10];
__device__ int mdq_gcn_trace_level;
#define GT_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_gcn_trace_buf[mdq_gcn_trace_level * 10 + (k)] += tn_ - tq_; tq_ = tn_; }
extern "C" int mdq_gcn_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_gcn_trace_buf), sizeof(long long) * 80) != hipSuccess) return -1;
  if (reset) { long long z[80] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_gcn_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define GT_STAMP(k)
#endif

struct Lds {
  float* x;      // level input features  [n][fin]      (row stride = fin)
  float* h;      // conv output           [n][C+1]
  float* agg;    // aggregated features   [n][fin]
  float* score;  // [n]
  float* deg;    // [n]  in-degree (+1 for GCN)
  int* adj_ptr;  // [n+1] CSR by target
  int* adj;      // [E]   source ids in edge order
  int* esrc;     // [E]
  int* edst;     // [E]
  int* newid;    // [n]
  int* misc;     // [8]
};

// CSR by target (sources kept in edge order => deterministic sums): in-degrees by LDS atomics, a workgroup-wide
// exclusive scan (wave shuffles + one exchange of the wave totals), an unordered fill of EDGE IDS through per-node
// cursors, then every node sorts its (short) list of edge ids and replaces them by the sources.  (The first version
// let 2-8 threads per node scan the whole edge list: O(n E) comparisons, 34 k of the kernel's 320 k cycles at level 0.)
__device__ __forceinline__ void build_csr(const Lds& L, int n, int E) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* cnt = L.newid;       // [n] degree, then fill cursor (newid is dead until this level's pooling)
  int* wtot = L.misc + 8;   // [WGT / 64] wave totals
  for (int i = tid; i < n; i += WGT) cnt[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += WGT) atomicAdd(&cnt[L.edst[e]], 1);
  __syncthreads();
  int carry = 0;
  for (int base = 0; base < n; base += WGT) {
    const int i = base + tid;
    const int c = i < n ? cnt[i] : 0;
    int v = c;   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    int pre = carry, all = carry;
    for (int w = 0; w < WGT / 64; ++w) {
      const int wt = wtot[w];
      if (w < wave) pre += wt;
      all += wt;
    }
    if (i < n) {
      L.adj_ptr[i] = pre + v - c;
      cnt[i] = 0;
    }
    carry = all;
    __syncthreads();   // (wtot is rewritten by the next pass)
  }
  if (tid == 0) L.adj_ptr[n] = carry;
  __syncthreads();
  for (int e = tid; e < E; e += WGT) {
    const int d = L.edst[e];
    L.adj[L.adj_ptr[d] + atomicAdd(&cnt[d], 1)] = e;
  }
  __syncthreads();
  for (int i = tid; i < n; i += WGT) {
    const int p0 = L.adj_ptr[i], p1 = L.adj_ptr[i + 1];
    for (int a_ = p0 + 1; a_ < p1; ++a_) {
      const int w = L.adj[a_];
      int j = a_ - 1;
      while (j >= p0 && L.adj[j] > w) {
        L.adj[j + 1] = L.adj[j];
        --j;
      }
      L.adj[j + 1] = w;
    }
    for (int q = p0; q < p1; ++q) L.adj[q] = L.esrc[L.adj[q]];
  }
  __syncthreads();
}

// out[i][c] = relu( b[c] + sum_f wl[f][c] * A[i][f] (+ wr[f][c] * X[i][f]) ) ;  A, X in LDS with stride fin
// Form (a): per node, weights of channel c in registers (fin <= 32).
// FIN > 0: the feature count is a compile-time constant (the reference's two models: 17 and 2) - the generic
// version spends most of its instructions on the 32 `f < fin` tests per node (measured: 52 of the kernel's 280 us).
template <bool ROOT, int FIN>
__device__ __forceinline__ void conv_dense_small_fin(const Lds& L, int n, int fin_rt, int C, const float* __restrict__ wl,
                                            const float* __restrict__ b, const float* __restrict__ wr,
                                            const float* A, const float* X) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // (keeps the per-row address arithmetic inside the level: see run_level)
  const int c = tid % C, g = tid / C, G = WGT / C;
  constexpr int NF = FIN > 0 ? FIN : 32;
  const int fin = FIN > 0 ? FIN : fin_rt;
  float wlr[NF], wrr[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    wlr[f] = f < fin ? wl[f * C + c] : 0.f;
    wrr[f] = (ROOT && f < fin) ? wr[f * C + c] : 0.f;
  }
  const float bc = b[c];
  for (int i = g; i < n; i += G) {
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (FIN > 0 || f < fin) {
        acc = fmaf(wlr[f], A[i * fin + f], acc);
        if (ROOT) acc = fmaf(wrr[f], X[i * fin + f], acc);
      }
    }
    L.h[i * (C + 1) + c] = acc + bc;
  }
}
template <bool ROOT>
__device__ __forceinline__ void conv_dense_small_fin_any(const Lds& L, int n, int fin, int C, const float* __restrict__ wl,
                                                const float* __restrict__ b, const float* __restrict__ wr,
                                                const float* A, const float* X) {
  if (fin == 17)
    conv_dense_small_fin<ROOT, 17>(L, n, fin, C, wl, b, wr, A, X);
  else if (fin == 2)
    conv_dense_small_fin<ROOT, 2>(L, n, fin, C, wl, b, wr, A, X);
  else
    conv_dense_small_fin<ROOT, 0>(L, n, fin, C, wl, b, wr, A, X);
}

// Form (b): feature-outer loop, up to NACC nodes per thread (n <= NACC * WGT/C), any fin.
// `wbuf` (LDS, fin*C floats, 16-byte aligned) or nullptr: when given, each weight matrix is first staged into LDS with
// one coalesced cooperative copy (all its loads in flight at once) - streaming the rows from global memory one
// dependent round trip per feature was 60 % of the whole kernel.
// NA = accumulators (nodes) per thread: the unrolled node loop costs its instructions whether a node exists or not,
// so the pooled levels (18, 2, 1 nodes) run instances with 5 / 1 accumulators; more than 8 rows per thread are served by
// repeated passes of the 8-row instance over row blocks `r0` (the 24-accumulator instance of rounds 2-3 could not be
// register-allocated without scratch; the weights of a level come from L2 either way).
template <bool ROOT, int NA>
__device__ __forceinline__ void conv_dense_small_n(const Lds& L, int n, int fin, int C, const float* __restrict__ wl,
                                          const float* __restrict__ b, const float* __restrict__ wr, const float* A,
                                          const float* X, float* wbuf, int r0 = 0) {
  constexpr int NACC = NA;   // (shadows the namespace constant inside this instance)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int c = tid % C, g = tid / C, G = WGT / C;
  float acc[NACC];
#pragma unroll
  for (int r = 0; r < NACC; ++r) acc[r] = 0.f;
  if (wbuf) {
    for (int idx = tid * 4; idx < fin * C; idx += WGT * 4)
      *reinterpret_cast<float4*>(wbuf + idx) = *reinterpret_cast<const float4*>(wl + idx);
    __syncthreads();
    for (int f = 0; f < fin; ++f) {
      const float w = wbuf[f * C + c];
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int i = g + (r0 + r) * G;
        if (i < n) acc[r] = fmaf(w, A[i * fin + f], acc[r]);
      }
    }
    if (ROOT) {
      __syncthreads();  // everybody is done with the first matrix
      for (int idx = tid * 4; idx < fin * C; idx += WGT * 4)
        *reinterpret_cast<float4*>(wbuf + idx) = *reinterpret_cast<const float4*>(wr + idx);
      __syncthreads();
      for (int f = 0; f < fin; ++f) {
        const float w = wbuf[f * C + c];
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          const int i = g + (r0 + r) * G;
          if (i < n) acc[r] = fmaf(w, X[i * fin + f], acc[r]);
        }
      }
    }
  } else if (g + r0 * G >= n) {
    // a row group without a row (n < G on the last pooled levels: 2 or 1 rows for 4 groups) streams no weights
  } else if ((fin & 63) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0 && (!ROOT || (reinterpret_cast<size_t>(X) & 15) == 0)) {
    // wide levels of the reference's widths (fin = C = 64 / 128 / 256): FBW features per block - FBW (x 2) weight loads in
    // flight per round trip instead of 16 (the pooled levels are a chain of global round trips: 8 -> 2 for C = 128 with
    // one node per thread) - and the operand rows read four features per LDS instruction.  Same fma order per output.
    constexpr int FBW = NACC <= 5 ? 32 : 8;   // (64 for one row: spilled beside the other instances)
    for (int f0 = 0; f0 < fin; f0 += FBW) {
      float w1[FBW], w2[FBW];
#pragma unroll
      for (int q = 0; q < FBW; ++q) {
        w1[q] = wl[(f0 + q) * C + c];
        w2[q] = ROOT ? wr[(f0 + q) * C + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < FBW; q += 4) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          const int i = min(g + (r0 + r) * G, n - 1);   // (rows past n re-read row n - 1, never stored)
          const float4 a4 = *reinterpret_cast<const float4*>(A + i * fin + f0 + q);
          float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ROOT) x4 = *reinterpret_cast<const float4*>(X + i * fin + f0 + q);
          acc[r] = fmaf(w1[q], a4.x, acc[r]);
          if (ROOT) acc[r] = fmaf(w2[q], x4.x, acc[r]);
          acc[r] = fmaf(w1[q + 1], a4.y, acc[r]);
          if (ROOT) acc[r] = fmaf(w2[q + 1], x4.y, acc[r]);
          acc[r] = fmaf(w1[q + 2], a4.z, acc[r]);
          if (ROOT) acc[r] = fmaf(w2[q + 2], x4.z, acc[r]);
          acc[r] = fmaf(w1[q + 3], a4.w, acc[r]);
          if (ROOT) acc[r] = fmaf(w2[q + 3], x4.w, acc[r]);
        }
      }
    }
  } else {
    constexpr int FB = NACC > 5 ? 8 : 16;
    // (issuing the next block's weight loads before the current block is used was measured slower: 18 -> 25 k cycles
    // on the one-node levels)
    for (int f0 = 0; f0 < fin; f0 += FB) {
      float w1[FB], w2[FB];
#pragma unroll
      for (int q = 0; q < FB; ++q) {
        const int f = f0 + q;
        w1[q] = f < fin ? wl[f * C + c] : 0.f;
        w2[q] = (ROOT && f < fin) ? wr[f * C + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < FB; ++q) {
        const int f = min(f0 + q, fin - 1);   // (features past fin: weight 0, any valid address)
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
          // rows past n re-read row n - 1 (never stored): unconditional reads, so that the LDS loads of a block
          // are in flight together instead of one conditional block - one LDS round trip - per (feature, row)
          const int i = min(g + (r0 + r) * G, n - 1);
          acc[r] = fmaf(w1[q], A[i * fin + f], acc[r]);
          if (ROOT) acc[r] = fmaf(w2[q], X[i * fin + f], acc[r]);
        }
      }
    }
  }
  const float bc = b[c];
  __syncthreads();  // (the staging buffer lives behind the rows of L.h written below; keep phases apart)
#pragma unroll
  for (int r = 0; r < NACC; ++r) {
    const int i = g + (r0 + r) * G;
    if (i < n) L.h[i * (C + 1) + c] = acc[r] + bc;
  }
}

// Form (c), C = 128 (16 channels per wave of the 512-thread workgroup) and enough nodes to fill lanes: lane = NODE, the
// wave's 16 channels in 16 accumulators per lane.  In forms (a) and (b) a wave's lanes are channels and every operand
// A[i][f] is ONE LDS address read by all 64 lanes - the LDS spends a full-rate read on a broadcast, and that traffic, not
// the FMAs, bounded the convolutions (level 0: 52 k cycles for 12 k cycles of FMA issue, level 1: 75 k).  Here a lane reads
// its own row (one LDS read per feature, re-used for 16 channels) and the weights W[f][c0 .. c0 + 15] are the same for the
// whole wave: read through the constant address space they arrive as ONE scalar load of 16 dwords per feature and feed
// the FMAs as scalar operands.  (The weights are not written inside this kernel; global and constant addresses coincide.)
// Same fma sequence per output as the other forms.
typedef float floatx16w __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) float* cfloatp;
typedef const __attribute__((address_space(4))) floatx16w* cfloat16p;
template <bool ROOT>
__device__ __forceinline__ void conv_dense_nodes(const Lds& L, int n, int fin, int C, const float* __restrict__ wl,
                                                 const float* __restrict__ b, const float* __restrict__ wr, const float* A,
                                                 const float* X) {
  const int lane = threadIdx.x & 63;
  const int c0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 16;
  const cfloatp wlc = (cfloatp)wl + c0, wrc = ROOT ? (cfloatp)wr + c0 : (cfloatp)wl + c0, bcp = (cfloatp)b + c0;
  const floatx16w bias = *(cfloat16p)bcp;
  for (int base = 0; base < n; base += 64) {
    const int i = min(base + lane, n - 1);      // (lanes past n re-read row n - 1, never stored)
    const float* ar = A + i * fin;
    const float* xr = X + i * fin;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll 2
    for (int f = 0; f < fin; ++f) {
      const floatx16w w1 = *(cfloat16p)(wlc + f * C);
      const float a = ar[f];
      floatx16w w2 = w1;
      float xv = 0.f;
      if (ROOT) {
        w2 = *(cfloat16p)(wrc + f * C);
        xv = xr[f];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k] = fmaf(w1[k], a, acc[k]);
        if (ROOT) acc[k] = fmaf(w2[k], xv, acc[k]);
      }
    }
    if (base + lane < n) {
#pragma unroll
      for (int k = 0; k < 16; ++k) L.h[i * (C + 1) + c0 + k] = acc[k] + bias[k];
    }
  }
}

// Form (d), the first POOLED level (8 < n <= 32 rows, fin = C): the convolution as 16 x 16 x 4 fp32 matrix-core products
// (with one or two rows the 16-row tiles are 90 % padding and the matrix pipe itself takes longer than form (b): 9.9 k
// against 7.6 k cycles measured).
// In forms (a) / (b) every operand A[i][f] is one LDS address read by a whole wave, and that broadcast traffic - 2 560
// wave-wide 16-byte reads for the 18 x 256 x 128 product of level 1 - bounded the level (37 k cycles for 10 k cycles of
// FMA issue).  v_mfma_f32_16x16x4_f32 IS a sequential fp32 fma chain over k (tools/micro/mfma_exact.hip: 0 of 51 200
// outputs differ from fmaf(a_k, b_k, acc) applied for k = 0, 1, 2, ... across instructions), so with chain position
// p = 2 f + (0: lin_l on the aggregate, 1: lin_r on the node's own row) as the k index every output is bitwise what the
// other forms compute.  The operand rows are staged once as T[i][p] behind the output rows of L.h with a row stride
// = 4 (mod 64): lane (row m, k-slot s) reads bank 4 m + s - conflict free.  Wave w owns the 16-column block(s)
// w, w + WGT / 64, ...: every weight is loaded exactly once per graph, KB k-steps of loads in flight.
typedef float v4f_t __attribute__((ext_vector_type(4)));
// floats of the staged operands of form (d): per 16-row block 64 rows (k-slot s, row m) of NK values + 4 of padding
__device__ __forceinline__ int mfma_stage_floats(int n, int KP) { return ((n + 15) >> 4) * 64 * ((KP >> 2) + 4); }
template <bool ROOT>
__device__ __forceinline__ void conv_dense_mfma(const Lds& L, int n, int fin, int C, const float* __restrict__ wl,
                                                const float* __restrict__ b, const float* __restrict__ wr, const float* A,
                                                const float* X) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int KP = ROOT ? 2 * fin : fin;          // chain positions per output
  const int NK = KP >> 2;                       // k-steps (instructions) per output block
  const int TS = NK + 4;                        // row stride of the staged operands: = 4 (mod 64) for NK = 64
  float* T = L.h + ((n * (C + 1) + 3) & ~3);
  const int lane = tid & 63, wave = tid >> 6, s = lane >> 4, m = lane & 15;
  const int wstep = ROOT ? 2 * C : 4 * C;
  constexpr int KB = 64;
  // the weights of this wave's first column block: requested before the staging below (they do not depend on it)
  float w[KB];
  {
    const int c0 = wave * 16;
    const float* wp = ROOT ? ((s & 1) ? wr : wl) + (s >> 1) * C + c0 + m : wl + s * C + c0 + m;
    if (wave < (C >> 4)) {
#pragma unroll
      for (int q = 0; q < KB; ++q) w[q] = (q < NK) ? wp[q * wstep] : 0.f;
    }
  }
  // operands, slot-major: T[(rb * 64 + s * 16 + m) * TS + t] = chain position p = 4 t + s of row rb * 16 + m, so that lane
  // (s, m) reads its k-steps t, t + 1, t + 2, t + 3 with ONE 16-byte LDS read (row index = lane id, stride = 4 mod 64:
  // conflict free) instead of four 4-byte reads in front of four dependent matrix instructions
  for (int idx = tid; idx < n * fin; idx += WGT) {
    const int i = idx / fin, f = idx - i * fin;
    float* tr = T + ((i >> 4) * 64 + (i & 15)) * TS;
    if (ROOT) {
      const int p = 2 * f;                                        // (p, p + 1: slots p & 3 and (p & 3) + 1, same t)
      tr[(p & 3) * 16 * TS + (p >> 2)] = A[idx];
      tr[((p & 3) + 1) * 16 * TS + (p >> 2)] = X[idx];
    } else {
      tr[(f & 3) * 16 * TS + (f >> 2)] = A[idx];
    }
  }
  __syncthreads();
  const bool two = n > 16;
  const float* t0p = T + (s * 16 + min(m, n - 1)) * TS;                       // (rows past n re-read row n - 1, never stored)
  const float* t1p = T + (64 + s * 16 + min(m, n - 17)) * TS;                 // (only read when n > 16)
  for (int cb = wave; cb < (C >> 4); cb += WGT / 64) {
    const int c0 = cb * 16;
    const float* wp = ROOT ? ((s & 1) ? wr : wl) + (s >> 1) * C + c0 + m : wl + s * C + c0 + m;
    v4f_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < NK; k0 += KB) {
      if (cb != wave || k0 != 0) {
#pragma unroll
        for (int q = 0; q < KB; ++q) w[q] = (k0 + q < NK) ? wp[(k0 + q) * wstep] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < KB; q += 4)
        if (k0 + q < NK) {                                   // (workgroup-uniform; NK is a multiple of 4 here)
          const float4 a0 = *reinterpret_cast<const float4*>(t0p + k0 + q);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w[q], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w[q + 1], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w[q + 2], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w[q + 3], acc0, 0, 0, 0);
          if (two) {
            const float4 a1 = *reinterpret_cast<const float4*>(t1p + k0 + q);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w[q], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w[q + 1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w[q + 2], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w[q + 3], acc1, 0, 0, 0);
          }
        }
    }
    const float bc = b[c0 + m];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i0 = s * 4 + r, i1 = 16 + s * 4 + r;
      if (i0 < n) L.h[i0 * (C + 1) + c0 + m] = acc0[r] + bc;
      if (two && i1 < n) L.h[i1 * (C + 1) + c0 + m] = acc1[r] + bc;
    }
  }
}

// level input buffer: level 0 holds [NMAX][fin0], later levels [ceil(ratio*NMAX)][C]
__host__ __device__ inline int mdq_gcn_xs(const mdq_gcn_net& net, int NMAX) {
  const int k1 = (int)ceil(net.ratio * (double)NMAX);
  const int a = NMAX * net.fin0, b = k1 * net.C;
  return ((a > b ? a : b) + 3) & ~3;
}

struct Level {
  int type, fin;
  const float *wl, *b, *wr, *pw;
};

// what the backward pass of one level needs from its forward pass (training kernel, mdq_gcn_train.hip): the rows of
// the k kept nodes only - TopKPooling passes a gradient to nothing else.  Pointers into the graph's global workspace.
struct TapeLevel {
  float* hsel;    // [k][C]   relu(conv) of kept node r
  float* ssel;    // [k]      its score
  float* aggsel;  // [k][fin] aggregated input features (the operand of lin_l / lin)
  float* xsel;    // [k][fin] its own input features (the operand of lin_r)
  int* perm;      // [k]      node id inside the level's input
  int* amax;      // [C]      kept node holding the channel maximum of the readout
  int* esrc;      // [E]      the level's input edges (nullptr: not wanted, level 0)
  int* edst;
};

// |pool_w| of every level, once per graph while the inputs are on their way: wave l walks level l's chain c = 0, 1, ...
// (the same fmaf chain every thread used to walk at every level - 3.2 k cycles of dependent FMAs on each level's
// critical path) and leaves the norm in L.misc[1 + l].  Needs nlevels <= 7 and the staging rows in L.h (free at kernel
// start); returns false when they do not fit - the levels then compute their norm themselves.
__device__ __forceinline__ bool pool_norms(const Lds& L, const mdq_gcn_net& net, int NMAX) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = net.C, CS = (C + 3) & ~3;
  if (net.nlevels > 7 || (size_t)NMAX * (C + 1) < (size_t)net.nlevels * CS) return false;
  for (int l = wave; l < net.nlevels; l += WGT / 64) {
    float* st = L.h + l * CS;
    const float* pw = net.levels[l].pool_w;
    for (int c = lane; c < C; c += 64) st[c] = pw[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float wn = 0.f;
#pragma unroll 16
    for (int c = 0; c < C; ++c) wn = fmaf(st[c], st[c], wn);
    if (lane == 0) reinterpret_cast<float*>(L.misc)[1 + l] = sqrtf(wn);
  }
  return true;
}

// one conv + relu + TopK pool + readout level; features in L.x ([n][fin]) are replaced by the pooled ones
// TAPE: the training kernel's instance (keeps the rows of the kept nodes); the inference instance carries none of it
// (One register allocation has to cover every form inlined here: rounds 2-3 carried 1.4 KB of scratch per lane - the
// address arithmetic of the 24-row form hoisted out of the level loop, spilled at kernel entry and re-read at every level
// by graphs that never ran that form.  Now: thread id opaque per level, at most 8 rows per pass, 32 weights in flight.)
template <bool TAPE = false>
__device__ __forceinline__ void run_level(const Lds& L, const Level& lv, int C, double ratio, int& n, int& E, int32_t* perm, float& rmax,
                                 float& rmean, int NMAX, const TapeLevel* tape = nullptr, float wn_pre = -1.f) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int fin = lv.fin;
#ifdef MDQ_GCN_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  // the pooling weights of this level, requested now and used after the convolution (as a load in front of the score
  // phase it was a global round trip - 3 k cycles - on the critical path of every level)
  float pwv[(256 + WGT - 1) / WGT];
#pragma unroll
  for (int q = 0; q < (256 + WGT - 1) / WGT; ++q) pwv[q] = (tid + q * WGT < C) ? lv.pw[tid + q * WGT] : 0.f;
  if (TAPE && tape->esrc)
    for (int e = tid; e < E; e += WGT) {
      tape->esrc[e] = L.esrc[e];
      tape->edst[e] = L.edst[e];
    }
  build_csr(L, n, E);
  GT_STAMP(0)
  // ---- aggregation into L.agg
  if (lv.type == 0) {  // SAGE: mean over incoming edges (duplicates count), 0 for isolated nodes
    for (int idx = tid; idx < n * fin; idx += WGT) {
      const int i = idx / fin, f = idx - i * fin;
      float s = 0.f;
      const int p0 = L.adj_ptr[i], p1 = L.adj_ptr[i + 1];
      for (int p = p0; p < p1; ++p) s += L.x[L.adj[p] * fin + f];
      L.agg[idx] = p1 > p0 ? s / (float)(p1 - p0) : 0.f;
    }
  } else {  // GCN: (A + I) with D^-1/2 on both sides, applied to X before the (linear) weight
    for (int i = tid; i < n; i += WGT) L.deg[i] = 1.0f / sqrtf((float)(L.adj_ptr[i + 1] - L.adj_ptr[i] + 1));
    __syncthreads();
    for (int idx = tid; idx < n * fin; idx += WGT) {
      const int i = idx / fin, f = idx - i * fin;
      const float di = L.deg[i];
      float s = di * di * L.x[idx];
      for (int p = L.adj_ptr[i]; p < L.adj_ptr[i + 1]; ++p) {
        const int j = L.adj[p];
        s += L.deg[j] * di * L.x[j * fin + f];
      }
      L.agg[idx] = s;
    }
  }
  __syncthreads();
  GT_STAMP(1)
  // ---- dense part
  const int G = WGT / C;
  // (form (c) reads 16 weights per scalar load: rows of 64-byte aligned segments - torch allocations are; other callers
  //  of the C ABI fall through to the other forms)
  const bool w64 = ((reinterpret_cast<size_t>(lv.wl) | reinterpret_cast<size_t>(lv.b) |
                     (lv.type == 0 ? reinterpret_cast<size_t>(lv.wr) : (size_t)0)) & 63) == 0;
  if (C == 16 * (WGT / 64) && n >= 16 && fin <= MDQ_GCN_FORMC_MAXFIN && w64) {        // lane = node, scalar weights (form (c))
    if (lv.type == 0)
      conv_dense_nodes<true>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x);
    else
      conv_dense_nodes<false>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x);
  } else if (fin <= 32) {
    if (lv.type == 0)
      conv_dense_small_fin_any<true>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x);
    else
      conv_dense_small_fin_any<false>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x);
  } else if (n > 8 && n <= 32 && (C & 15) == 0 && (fin & (lv.type == 0 ? 7 : 15)) == 0 && (reinterpret_cast<size_t>(L.h) & 15) == 0 &&
             ((n * (C + 1) + 3) & ~3) + mfma_stage_floats(n, lv.type == 0 ? 2 * fin : fin) <= NMAX * (C + 1)) {   // matrix cores (form (d))
    if (lv.type == 0)
      conv_dense_mfma<true>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x);
    else
      conv_dense_mfma<false>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x);
  } else {
    // (n <= NACC*G is guaranteed by the host-side check)
    // (staging each weight matrix in LDS behind the rows of L.h - conv_dense_small_n's `wbuf` path - was measured
    // SLOWER than the blocked global loads, 0.50 vs 0.33 ms for 128 graphs, so it stays off)
    (void)NMAX;
    float* wbuf = nullptr;
    const int per = (n + G - 1) / G;   // nodes per thread (workgroup-uniform)
    if (lv.type == 0) {
      if (per <= 1) conv_dense_small_n<true, 1>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x, wbuf);
      else if (per <= 5) conv_dense_small_n<true, 5>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x, wbuf);
      else if (per <= 8) conv_dense_small_n<true, 8>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x, wbuf);
      else
        for (int r0 = 0; r0 < per; r0 += 8)
          conv_dense_small_n<true, 8>(L, n, fin, C, lv.wl, lv.b, lv.wr, L.agg, L.x, wbuf, r0);
    } else {
      if (per <= 1) conv_dense_small_n<false, 1>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x, wbuf);
      else if (per <= 5) conv_dense_small_n<false, 5>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x, wbuf);
      else if (per <= 8) conv_dense_small_n<false, 8>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x, wbuf);
      else
        for (int r0 = 0; r0 < per; r0 += 8)
          conv_dense_small_n<false, 8>(L, n, fin, C, lv.wl, lv.b, nullptr, L.agg, L.x, wbuf, r0);
    }
  }
  (void)G;
  __syncthreads();
  GT_STAMP(2)
  // ---- relu + score = tanh(h . w / |w|)
  // pool weights once into LDS (L.deg is free here: the aggregation is done)
#pragma unroll
  for (int q = 0; q < (256 + WGT - 1) / WGT; ++q)
    if (tid + q * WGT < C) L.deg[tid + q * WGT] = pwv[q];
  __syncthreads();
  // |w|: the same chain c = 0, 1, ... in every thread that needs it (one 16-byte LDS read per four channels; waves
  // without a node skip it - on the pooled levels that is all but the first, and the LDS pipe is what this costs)
  float wn = wn_pre;
  if (wn_pre < 0.f && (tid & ~63) / 16 < n) {
    wn = 0.f;
    if ((C & 3) == 0 && (reinterpret_cast<size_t>(L.deg) & 15) == 0) {
#pragma unroll 8
      for (int c = 0; c < C; c += 4) {
        const float4 d4 = *reinterpret_cast<const float4*>(L.deg + c);
        wn = fmaf(d4.x, d4.x, wn);
        wn = fmaf(d4.y, d4.y, wn);
        wn = fmaf(d4.z, d4.z, wn);
        wn = fmaf(d4.w, d4.w, wn);
      }
    } else {
#pragma unroll 16
      for (int c = 0; c < C; ++c) wn = fmaf(L.deg[c], L.deg[c], wn);
    }
    wn = sqrtf(wn);
  }
  GT_STAMP(8)
  {
    // 16 lanes per node, each lane a strided slice of the channels, butterfly sum (relu in place + dot product with the
    // pooling weights); WGT / 16 nodes per pass.  (One thread per node walking all channels - the first version of the
    // many-node case - kept 3 of 8 waves busy on a 128-step dependent chain: 22 k cycles at level 0.)
    const int sub = tid % 16;
    for (int base = 0; base < n; base += WGT / 16) {
      const int i = base + tid / 16;
      float sp = 0.f;
      if (i < n)
        for (int c = sub; c < C; c += 16) {
          float hv = L.h[i * (C + 1) + c];
          hv = hv > 0.f ? hv : 0.f;
          L.h[i * (C + 1) + c] = hv;
          sp = fmaf(hv, L.deg[c], sp);
        }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sp += __shfl_xor(sp, off, 16);
      if (i < n && sub == 0) L.score[i] = tanhf(sp / wn);
    }
  }
  __syncthreads();
  GT_STAMP(3)
  // ---- top-k by rank counting (descending score, ties by lower index = stable sort)
  // two lanes per node, each counting over half of the scores (integer partial ranks: the sum is exact in any order),
  // four scores per LDS read where the array is 16-byte aligned; kept[r] = the node of rank r < k (L.adj_ptr is free:
  // the aggregation is done) for the pooled-feature loop below
  const int k = (int)ceil(ratio * (double)n);
  int* kept = L.adj_ptr;
  {
    const bool s4 = (reinterpret_cast<size_t>(L.score) & 15) == 0;
    const int jm = min(((((n + 1) >> 1) + 3) & ~3), n), half = tid & 1;
    for (int base = 0; base < n; base += WGT / 2) {
      const int i = base + (tid >> 1);
      int rank = 0;
      if (i < n) {
        const float si = L.score[i];
        int j = half ? jm : 0;
        const int j1 = half ? n : jm;
        if (s4)
          for (; j + 4 <= j1; j += 4) {
            const float4 q = *reinterpret_cast<const float4*>(L.score + j);
            rank += (q.x > si) || (q.x == si && j < i);
            rank += (q.y > si) || (q.y == si && j + 1 < i);
            rank += (q.z > si) || (q.z == si && j + 2 < i);
            rank += (q.w > si) || (q.w == si && j + 3 < i);
          }
        for (; j < j1; ++j) {
          const float sj = L.score[j];
          rank += (sj > si) || (sj == si && j < i);
        }
      }
      rank += __shfl_xor(rank, 1);
      if (i < n && half == 0) {
        L.newid[i] = rank < k ? rank : -1;
        if (rank < k) kept[rank] = i;
        if (perm) {             // TopKPooling's `perm` (kept node r of this level = node perm[r] of its input), on request
          if (rank < k) perm[rank] = i;
          if (i >= k) perm[i] = -1;
        }
      }
    }
  }
  __syncthreads();
  GT_STAMP(4)
  if (TAPE) {   // rows of the kept nodes, before the pooled features overwrite the level input
    for (int idx = tid; idx < n * C; idx += WGT) {
      const int i = idx / C, c = idx - i * C, r = L.newid[i];
      if (r >= 0) tape->hsel[r * C + c] = L.h[i * (C + 1) + c];
    }
    for (int idx = tid; idx < n * fin; idx += WGT) {
      const int i = idx / fin, f = idx - i * fin, r = L.newid[i];
      if (r >= 0) {
        tape->aggsel[r * fin + f] = L.agg[idx];
        tape->xsel[r * fin + f] = L.x[idx];
      }
    }
    for (int i = tid; i < n; i += WGT) {
      const int r = L.newid[i];
      if (r >= 0) {
        tape->ssel[r] = L.score[i];
        tape->perm[r] = i;
      }
    }
    __syncthreads();
  }
  // ---- pooled features x'[r][c] = h[perm r][c] * score[perm r]  -> L.x with stride C
  // (NaN scores - a diverged network - rank nothing consistently: ranks are then no permutation and some kept[r] still hold
  //  adjacency pointers of the aggregation phase; the clamp keeps every read inside the level's arrays, the NaNs propagate)
  for (int idx = tid; idx < k * C; idx += WGT) {
    const int r = idx / C, c = idx - r * C;
    const int i = min(max(kept[r], 0), n - 1);
    L.x[idx] = L.h[i * (C + 1) + c] * L.score[i];
  }
  GT_STAMP(5)
  // ---- filter + relabel edges, preserving edge order (wave 0, ballot compaction)
  if (tid < 64) {
    int outp = 0;
    for (int base = 0; base < E; base += 64) {
      const int e = base + tid;
      int s = -1, d = -1;
      if (e < E) {
        s = L.newid[L.esrc[e]];
        d = L.newid[L.edst[e]];
      }
      const bool keep = (s >= 0) && (d >= 0);
      const unsigned long long m = __ballot(keep);
      const int pos = outp + __popcll(m & ((1ull << tid) - 1ull));
      // (in-place is safe: pos <= e, and all reads of this chunk happened before the ballot)
      if (keep) {
        L.esrc[pos] = s;
        L.edst[pos] = d;
      }
      outp += __popcll(m);
    }
    if (tid == 0) L.misc[0] = outp;
  }
  __syncthreads();
  GT_STAMP(6)
  E = L.misc[0];
  n = k;
  // ---- readout over the pooled nodes: thread c < C
  if (tid < C) {
    float mx = -INFINITY, sm = 0.f;
    int arg = 0;
    for (int r = 0; r < n; ++r) {
      const float v = L.x[r * C + tid];
      if (TAPE && v > mx) arg = r;      // first maximum (where torch's max sends the gradient)
      mx = fmaxf(mx, v);
      sm += v;
    }
    rmax += mx;
    rmean += sm / (float)n;
    if (TAPE) tape->amax[tid] = arg;
  }
  __syncthreads();
  GT_STAMP(7)
}

__global__ __launch_bounds__(WGT) void gcn_embed_kernel(mdq_gcn_net net, int NMAX, int EMAX, const float* x,
                                                         const int32_t* node_ptr, const int32_t* esrc,
                                                         const int32_t* edst, const int32_t* edge_ptr,
                                                         const int32_t* edge_cnt, float* emb, int32_t* perm,
                                                         int32_t* status) {
  extern __shared__ __align__(16) float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, C = net.C;
  const int n0 = node_ptr[b], nn = node_ptr[b + 1] - n0;
  // edge lists: packed (offsets edge_ptr) or padded to EMAX slots per graph with the counts beside them (edge_cnt)
  const int e0 = edge_cnt ? b * EMAX : edge_ptr[b], ne = edge_cnt ? edge_cnt[b] : edge_ptr[b + 1] - e0;
  if (status && tid == 0) status[b] = (nn > NMAX || nn < 0) ? -1 : (ne > EMAX || ne < 0) ? -2 : 0;
  if (nn > NMAX || ne > EMAX || nn < 0 || ne < 0) {
    // the LDS carve-up is sized from NMAX / EMAX: a larger graph must not be staged.  Its outputs are NaN (never a
    // plausible Q-value) and, where the caller passed one, its status says why
    if (tid < 2 * C) emb[(size_t)b * 2 * C + tid] = __builtin_nanf("");
    return;
  }
  const int XS = mdq_gcn_xs(net, NMAX);  // floats of the level-input / aggregation buffers
  Lds L;
  float* p = sm;
  L.x = p;      p += XS;
  L.h = p;      p += (size_t)NMAX * (C + 1);
  L.agg = p;    p += XS;
  L.score = p;  p += NMAX;
  L.deg = p;    p += (NMAX > C ? NMAX : C);  // (also stages the C pooling weights)
  // integer arrays: 16-byte aligned, edge arrays padded to whole quads (build_csr reads them as int4)
  int* q = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
  const int EQ = (EMAX + 3) & ~3;
  L.adj_ptr = q; q += (NMAX + 1 + 3) & ~3;
  L.adj = q;     q += EQ;
  L.esrc = q;    q += EQ;
  L.edst = q;    q += EQ;
  L.newid = q;   q += NMAX;
  L.misc = q;    // [8 + WGT / 64]
  for (int idx = tid; idx < nn * net.fin0; idx += WGT) L.x[idx] = x[(size_t)n0 * net.fin0 + idx];
  for (int e = tid; e < ne; e += WGT) {
    L.esrc[e] = esrc[e0 + e];
    L.edst[e] = edst[e0 + e];
  }
  const bool norms = pool_norms(L, net, NMAX);
  __syncthreads();
  int n = nn, E = ne;
  float rmax = 0.f, rmean = 0.f;
  for (int l = 0; l < net.nlevels; ++l) {
    Level lv;
    lv.type = net.levels[l].type;
    lv.fin = net.levels[l].fin;
    lv.wl = net.levels[l].w_l;
    lv.b = net.levels[l].b;
    lv.wr = net.levels[l].w_r;
    lv.pw = net.levels[l].pool_w;
#ifdef MDQ_GCN_TRACE
    if (tid == 0 && b == 0) mdq_gcn_trace_level = l;
    __syncthreads();
#endif
    run_level(L, lv, C, net.ratio, n, E, perm ? perm + ((size_t)b * net.nlevels + l) * NMAX : nullptr, rmax, rmean, NMAX, nullptr,
              norms ? reinterpret_cast<const float*>(L.misc)[1 + l] : -1.f);
  }
  if (tid < C) {
    emb[(size_t)b * 2 * C + tid] = rmax;
    emb[(size_t)b * 2 * C + C + tid] = rmean;
  }
}

// ------------------------------------------------------------------ MLP head on the matrix cores
typedef float floatx16 __attribute__((ext_vector_type(16)));

// OUT[32][ncols] = act( IN[32][K] * W[K][ncols] + bias ), IN / OUT in LDS (row-major), W in global
// (row-major [K][ncols] = transposed torch weight).  4 waves, one 32x32 block each per pass.
__device__ inline void head_layer(const float* in, int in_stride, int K, const float* __restrict__ W,
                                  const float* __restrict__ bias, int ncols, float* out, int out_stride, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = (ncols + 31) / 32;
  for (int blk = wave; blk < nblk; blk += WGH / 64) {
    const int col = blk * 32 + (lane & 31);
    const bool cv = col < ncols;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kh = lane >> 5;
    // KU steps of the k loop at a time: their weight loads are independent and in flight together (one global
    // round trip per step made the three layers 78 us for 128 graphs, all of it load latency)
    constexpr int KU = 16;
    for (int k0 = 0; k0 < K; k0 += 2 * KU) {
      float bv[KU], a[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + 2 * u + kh;
        bv[u] = (cv && k < K) ? W[(size_t)k * ncols + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + 2 * u + kh;
        a[u] = k < K ? in[(lane & 31) * in_stride + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u], acc, 0, 0, 0);
    }
    if (cv) {
      const float bc = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r] + bc;
        if (relu) v = v > 0.f ? v : 0.f;
        out[row * out_stride + col] = v;
      }
    }
  }
}

__device__ __forceinline__ void head_store(const mdq_gcn_net& net, int B, int g0, const float* a3, int OUTP, float* out);

__global__ __launch_bounds__(WGH) void mlp_head_kernel(mdq_gcn_net net, int B, const float* emb, float* out) {
  extern __shared__ __align__(16) float sm[];
  const int tid = threadIdx.x, K1 = 2 * net.C, OUT = net.out_dim;
  const int OUTP = (OUT + 31) & ~31;
  // row strides padded by one float: the A-operand reads walk 32 rows at a fixed k (bank conflicts otherwise)
  float* a0 = sm;                    // [32][K1+1]
  float* a1 = a0 + 32 * (K1 + 1);    // [32][129]
  float* a2 = a1 + 32 * 129;         // [32][65]
  float* a3 = a2 + 32 * 65;          // [32][OUTP]
  const int g0 = blockIdx.x * 32;
  for (int idx = tid; idx < 32 * K1; idx += WGH) {
    const int r = idx / K1, c = idx - r * K1;
    a0[r * (K1 + 1) + c] = (g0 + r < B) ? emb[(size_t)(g0 + r) * K1 + c] : 0.f;
  }
  __syncthreads();
  head_layer(a0, K1 + 1, K1, net.lin1_w, net.lin1_b, 128, a1, 129, true);
  __syncthreads();
  head_layer(a1, 129, 128, net.lin2_w, net.lin2_b, 64, a2, 65, true);
  __syncthreads();
  head_layer(a2, 65, 64, net.lin3_w, net.lin3_b, OUT, a3, OUTP, false);
  __syncthreads();
  head_store(net, B, g0, a3, OUTP, out);
}

// The same head for the reference's width (C = 128: 256 -> 128 -> 64 -> out_dim <= 256) with EVERY weight of a wave's
// blocks loaded up front - lin1 128, lin2 64, lin3 2 x 32 values per lane, none of which depends on an activation - so
// the kernel pays one burst of global round trips beside the staging of the embeddings instead of sixteen dependent
// ones (measured per 128 graphs: 42 us alone, 90 us while the flow leg's set-up kernel keeps the memory system busy,
// almost all of it load latency).  Same MFMA sequence per output as `head_layer`: bitwise the same results.
template <int KH>
__device__ __forceinline__ void head_wload(const float* __restrict__ W, int ncols, int blk, float (&bv)[KH]) {
  const int lane = threadIdx.x & 63, kh = lane >> 5;
  const int col = min(blk * 32 + (lane & 31), ncols - 1);   // (columns past ncols: any valid address, never stored)
#pragma unroll
  for (int u = 0; u < KH; ++u) bv[u] = W[(size_t)(2 * u + kh) * ncols + col];
}

template <int KH>
__device__ __forceinline__ void head_mma(const float* in, int in_stride, const float (&bv)[KH], float bc, int ncols, int blk,
                                         float* out, int out_stride, bool relu) {
  const int lane = threadIdx.x & 63, kh = lane >> 5, col = blk * 32 + (lane & 31);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int KU = 16;
#pragma unroll
  for (int u0 = 0; u0 < KH; u0 += KU) {
    float a[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) a[u] = in[(lane & 31) * in_stride + 2 * (u0 + u) + kh];
#pragma unroll
    for (int u = 0; u < KU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u0 + u], acc, 0, 0, 0);
  }
  if (col < ncols) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float v = acc[r] + bc;
      if (relu) v = v > 0.f ? v : 0.f;
      out[row * out_stride + col] = v;
    }
  }
}

__device__ __forceinline__ void head_store(const mdq_gcn_net& net, int B, int g0, const float* a3, int OUTP, float* out) {
  const int tid = threadIdx.x, OUT = net.out_dim;
  const int lane = tid & 63, wave = tid >> 6;
  for (int r = wave; r < 32; r += WGH / 64) {
    if (g0 + r >= B) continue;
    const float* row = a3 + r * OUTP;
    if (net.softmax) {
      float mx = -INFINITY;
      for (int c = lane; c < OUT; c += 64) mx = fmaxf(mx, row[c]);
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float s = 0.f;
      for (int c = lane; c < OUT; c += 64) s += expf(row[c] - mx);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      for (int c = lane; c < OUT; c += 64) out[(size_t)(g0 + r) * OUT + c] = expf(row[c] - mx) / s;
    } else {
      for (int c = lane; c < OUT; c += 64) out[(size_t)(g0 + r) * OUT + c] = row[c];
    }
  }
}

__global__ __launch_bounds__(WGH) void mlp_head_c128_kernel(mdq_gcn_net net, int B, const float* emb, float* out) {
  extern __shared__ __align__(16) float sm[];
  constexpr int K1 = 256;
  const int tid = threadIdx.x, OUT = net.out_dim, lane = tid & 63, wave = tid >> 6;
  const int OUTP = (OUT + 31) & ~31, nblk3 = OUTP / 32;
  float* a0 = sm;                    // [32][K1+1]
  float* a1 = a0 + 32 * (K1 + 1);    // [32][129]
  float* a2 = a1 + 32 * 129;         // [32][65]
  float* a3 = a2 + 32 * 65;          // [32][OUTP]
  const int g0 = blockIdx.x * 32;
  // every weight this wave will use (wave-uniform branches)
  float w1[128], w2[64], w3a[32], w3b[32];
  float b1, b2 = 0.f, b3a = 0.f, b3b = 0.f;
  head_wload<128>(net.lin1_w, 128, wave, w1);
  b1 = net.lin1_b[wave * 32 + (lane & 31)];
  if (wave < 2) {
    head_wload<64>(net.lin2_w, 64, wave, w2);
    b2 = net.lin2_b[wave * 32 + (lane & 31)];
  }
  if (wave < nblk3) {
    head_wload<32>(net.lin3_w, OUT, wave, w3a);
    b3a = net.lin3_b[min(wave * 32 + (lane & 31), OUT - 1)];
  }
  if (wave + 4 < nblk3) {
    head_wload<32>(net.lin3_w, OUT, wave + 4, w3b);
    b3b = net.lin3_b[min((wave + 4) * 32 + (lane & 31), OUT - 1)];
  }
  for (int idx = tid; idx < 32 * K1; idx += WGH) {
    const int r = idx / K1, c = idx - r * K1;
    a0[r * (K1 + 1) + c] = (g0 + r < B) ? emb[(size_t)(g0 + r) * K1 + c] : 0.f;
  }
  __syncthreads();
  head_mma<128>(a0, K1 + 1, w1, b1, 128, wave, a1, 129, true);
  __syncthreads();
  if (wave < 2) head_mma<64>(a1, 129, w2, b2, 64, wave, a2, 65, true);
  __syncthreads();
  if (wave < nblk3) head_mma<32>(a2, 65, w3a, b3a, OUT, wave, a3, OUTP, false);
  if (wave + 4 < nblk3) head_mma<32>(a2, 65, w3b, b3b, OUT, wave + 4, a3, OUTP, false);
  __syncthreads();
  head_store(net, B, g0, a3, OUTP, out);
}

}  // namespace mdq_gcn

extern "C" int mdq_gcn_forward(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                               const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst,
                               const int32_t* edge_ptr, float* emb, float* out, void* stream) {
  return mdq_gcn_forward_ex(net, B, NMAX, EMAX, x, node_ptr, esrc, edst, edge_ptr, emb, out, nullptr, nullptr, stream);
}

static int gcn_forward_impl(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                            const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst, const int32_t* edge_ptr,
                            const int32_t* edge_cnt, float* emb, float* out, int32_t* perm, int32_t* status, void* stream);

extern "C" int mdq_gcn_forward_ex(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                                  const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst,
                                  const int32_t* edge_ptr, float* emb, float* out, int32_t* perm, int32_t* status,
                                  void* stream) {
  if (!edge_ptr) return mdq_set_error("mdq_gcn_forward: bad arguments");
  return gcn_forward_impl(net, B, NMAX, EMAX, x, node_ptr, esrc, edst, edge_ptr, nullptr, emb, out, perm, status, stream);
}

extern "C" int mdq_gcn_forward_padded(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                                      const int32_t* node_ptr, const int32_t* esrc_pad, const int32_t* edst_pad,
                                      const int32_t* edge_cnt, float* emb, float* out, int32_t* perm, int32_t* status,
                                      void* stream) {
  if (!edge_cnt) return mdq_set_error("mdq_gcn_forward_padded: bad arguments");
  return gcn_forward_impl(net, B, NMAX, EMAX, x, node_ptr, esrc_pad, edst_pad, nullptr, edge_cnt, emb, out, perm, status, stream);
}

static int gcn_forward_impl(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                            const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst, const int32_t* edge_ptr,
                            const int32_t* edge_cnt, float* emb, float* out, int32_t* perm, int32_t* status, void* stream) {
  using namespace mdq_gcn;
  if (!net || B <= 0 || !x || !node_ptr || (!edge_ptr && !edge_cnt) || !emb || !out) return mdq_set_error("mdq_gcn_forward: bad arguments");
  const int C = net->C;
  if (C != 64 && C != 128 && C != 256 && C != 32) return mdq_set_error("mdq_gcn_forward: conv width must divide 256");
  if (net->fin0 > 32 && NMAX > NACC * (WGT / C)) return mdq_set_error("mdq_gcn_forward: graph too large for the input width");
  for (int l = 0; l < net->nlevels; ++l) {
    // node count entering level l
    double n = NMAX;
    for (int j = 0; j < l; ++j) n = std::ceil(net->ratio * n);
    if (net->levels[l].fin > 32 && n > NACC * (WGT / C)) return mdq_set_error("mdq_gcn_forward: too many nodes at a wide level");
  }
  size_t lds = sizeof(float) * ((size_t)mdq_gcn_xs(*net, NMAX) * 2 + (size_t)NMAX * (C + 1) + (size_t)NMAX +
                                (size_t)(NMAX > C ? NMAX : C)) +
               sizeof(int) * ((size_t)NMAX + 4 + 3 * ((size_t)EMAX + 3) + NMAX + 8 + WGT / 64) + 16;
  if (lds > 160 * 1024) return mdq_set_error("mdq_gcn_forward: graph does not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_embed_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  hipLaunchKernelGGL(gcn_embed_kernel, dim3(B), dim3(WGT), lds, st, *net, NMAX, EMAX, x, node_ptr, esrc, edst,
                     edge_ptr, edge_cnt, emb, perm, status);
  e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  const int OUTP = (net->out_dim + 31) & ~31;
  size_t lds2 = sizeof(float) * 32 * ((size_t)2 * C + 1 + 129 + 65 + OUTP);
  const bool c128 = C == 128 && net->out_dim <= 256;   // the reference's width: all weights of a wave loaded up front
  e = hipFuncSetAttribute(c128 ? reinterpret_cast<const void*>(&mlp_head_c128_kernel) : reinterpret_cast<const void*>(&mlp_head_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  if (c128)
    hipLaunchKernelGGL(mlp_head_c128_kernel, dim3((B + 31) / 32), dim3(WGH), lds2, st, *net, B, emb, out);
  else
    hipLaunchKernelGGL(mlp_head_kernel, dim3((B + 31) / 32), dim3(WGH), lds2, st, *net, B, emb, out);
  e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  return 0;
}
