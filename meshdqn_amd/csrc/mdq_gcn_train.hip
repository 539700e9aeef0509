// The learning step of the graph Q-network on gfx950: forward, double-DQN Huber loss and the complete backward pass
// of `NodeRemovalNet` / `AirfoilGCNN` for a minibatch of graphs, hand-written (no autograd, no dense adjacency).
//
// Replaces `DataWorker.compute_gradients` (airfoil_dqn.py:240-310: loss.backward() through PyG's SAGEConv / GCNConv /
// TopKPooling / global max + mean pool and the three Linear layers) for the network that is being trained.
//
// Kernel 1 (gcn_train_kernel): ONE 512-thread workgroup per graph of the minibatch.  The loss is a mean over the
// graphs, so each graph's gradient is independent of the others: the workgroup runs the forward levels out of LDS
// exactly like the inference kernel (same code: run_level), keeping on a small tape only the ROWS OF THE KEPT NODES -
// TopKPooling hands a gradient to nothing else, so with ratio 0.1 the backward pass of the 180-node level touches 18
// rows - then the head forward, the loss term of the graph, the head backward and the levels backwards:
//     readout      d x'[r][c]  = g_mean[c] / k + [r == argmax_c] g_max[c]  (+ what the next level passes down)
//     TopKPooling  x' = h[perm] * s,  s = tanh(h . w / |w|):  d h, d w  (segmented over the kept rows)
//     relu, SAGEConv / GCNConv: weight gradients = (kept rows of d pre)^T (kept rows of agg | x): a skinny GEMM per
//     level; input gradients = d pre W scattered back along the level's edges in edge order (segmented scatter-add,
//     one thread per feature: deterministic sums).
// Every parameter gradient of the graph goes to the graph's own slice of `partial` (torch layout, where the flat
// gradient of the all-reduce wants it).
// Kernel 2 (gcn_grad_reduce_kernel): flat gradient = sum of the slices in graph order, loss = mean of the terms:
// bitwise reproducible, no atomics anywhere.
//
// Kernel 0 (gcn_pack_kernel): the parameters of the module (torch layout [out][in]) into the kernels' layout
// ([in][out]: consecutive lanes read consecutive output channels) in ONE launch after every optimiser step.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_gcn {

#ifdef MDQ_GCN_TRAIN_PROF
// debug build only: s_memtime at the phase boundaries of graph 0 (mdq_gcn_train_prof_host)
__device__ long long mdq_gcn_train_prof[32];
#define TP_STAMP(k) { if (blockIdx.x == 0 && threadIdx.x == 0) mdq_gcn_train_prof[k] = __builtin_amdgcn_s_memtime(); }
extern "C" int mdq_gcn_train_prof_host(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_gcn_train_prof), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
#else
#define TP_STAMP(k)
#endif

struct TapeOffsets {
  int hsel, ssel, aggsel, xsel, perm, amax, esrc, edst;   // in 4-byte units from the graph's workspace; esrc < 0: none
};

__device__ __forceinline__ TapeLevel tape_level(float* ws, const TapeOffsets& o) {
  TapeLevel t;
  t.hsel = ws + o.hsel;
  t.ssel = ws + o.ssel;
  t.aggsel = ws + o.aggsel;
  t.xsel = ws + o.xsel;
  t.perm = reinterpret_cast<int*>(ws + o.perm);
  t.amax = reinterpret_cast<int*>(ws + o.amax);
  t.esrc = o.esrc >= 0 ? reinterpret_cast<int*>(ws + o.esrc) : nullptr;
  t.edst = o.edst >= 0 ? reinterpret_cast<int*>(ws + o.edst) : nullptr;
  return t;
}

// workspace of one graph: [0] its loss term, then the tape of every level sized for NMAX nodes / EMAX edges
__host__ __device__ inline int tape_layout(const mdq_gcn_net& net, int NMAX, int EMAX, TapeOffsets* off) {
  int pos = 4, n = NMAX;
  for (int l = 0; l < net.nlevels; ++l) {
    const int k = (int)ceil(net.ratio * (double)n), fin = net.levels[l].fin, C = net.C;
    TapeOffsets o;
    o.hsel = pos;   pos += k * C;
    o.ssel = pos;   pos += k;
    o.aggsel = pos; pos += k * fin;
    o.xsel = pos;   pos += k * fin;
    o.perm = pos;   pos += k;
    o.amax = pos;   pos += C;
    o.esrc = o.edst = -1;
    if (l > 0) {
      o.esrc = pos; pos += EMAX;
      o.edst = pos; pos += EMAX;
    }
    if (off) off[l] = o;
    n = k;
  }
  return (pos + 3) & ~3;
}

// pre[j] = b[j] + sum_k in[k] W[k][j]  (W in the kernels' layout [K][N], N <= 256); the k range is split over the
// WGT / NP thread groups, the partial sums are added in group order
__device__ inline void head_fwd(const float* in, int K, const float* __restrict__ W, const float* __restrict__ bias, int N,
                                float* out, float* scratch) {
  const int tid = threadIdx.x;
  int NP = 64;
  while (NP < N) NP <<= 1;
  const int P = WGT / NP, j = tid % NP, p = tid / NP;
  const int kq = (K + P - 1) / P, k0 = p * kq, k1 = min(K, k0 + kq);
  float acc = 0.f;
  if (j < N) {
#pragma unroll 8
    for (int k = k0; k < k1; ++k) acc = fmaf(in[k], W[(size_t)k * N + j], acc);
  }
  scratch[p * NP + j] = acc;
  __syncthreads();
  if (tid < N) {
    float s = bias[tid];
    for (int q = 0; q < P; ++q) s += scratch[q * NP + tid];
    out[tid] = s;
  }
  __syncthreads();
}

// din[k] = sum_j dz[j] W[k][j]: one wave per row k, lanes over j (coalesced), butterfly sum
__device__ inline void head_bwd_in(const float* dz, int N, const float* __restrict__ W, int K, float* din) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < K; k += WGT / 64) {
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s = fmaf(dz[j], W[(size_t)k * N + j], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) din[k] = s;
  }
  __syncthreads();
}

// gW[j][k] = dz[j] a[k] (torch layout [N][K]), gb = dz
__device__ inline void head_bwd_w(const float* dz, int N, const float* a, int K, float* gW, float* gb) {
  for (int idx = threadIdx.x; idx < N * K; idx += WGT) {
    const int j = idx / K, k = idx - j * K;
    gW[idx] = dz[j] * a[k];
  }
  if ((int)threadIdx.x < N) gb[threadIdx.x] = dz[threadIdx.x];
}

__global__ __launch_bounds__(WGT) void gcn_train_kernel(mdq_gcn_net net, mdq_gcn_train_desc D, int wstride) {
  extern __shared__ __align__(16) float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, C = net.C, OUT = net.out_dim, NMAX = D.NMAX, EMAX = D.EMAX;
  const int n0 = D.node_ptr[b], nn = D.node_ptr[b + 1] - n0;
  const int e0 = D.edge_ptr[b], ne = D.edge_ptr[b + 1] - e0;
  float* ws = D.workspace + (size_t)b * wstride;
  float* gp = D.partial + (size_t)b * D.layout.total;
  if (nn > NMAX || ne > EMAX || nn <= 0 || ne < 0) {
    // a graph the LDS carve-up was not sized for: its loss term is NaN (so is the loss), it adds no gradient (its slice
    // still holds the previous minibatch's values: cleared)
    if (tid == 0) ws[0] = __builtin_nanf("");
    for (int i = tid; i < D.layout.total; i += WGT) gp[i] = 0.f;
    return;
  }
  const int XS = mdq_gcn_xs(net, NMAX);
  const int KM = (int)ceil(net.ratio * (double)NMAX);   // kept nodes of the first level: the most of any level
  Lds L;
  float* p = sm;
  L.x = p;      p += XS;
  L.h = p;      p += (size_t)NMAX * (C + 1);
  L.agg = p;    p += XS;
  L.score = p;  p += NMAX;
  L.deg = p;    p += (NMAX > C ? NMAX : C);
  float* gemb = p;  p += 2 * C;           // d loss / d embedding
  float* rawb = p;  p += (KM + 3) & ~3;   // pre-tanh scores of the kept nodes
  int* q = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
  const int EQ = (EMAX + 3) & ~3;
  L.adj_ptr = q; q += (NMAX + 1 + 3) & ~3;
  L.adj = q;     q += EQ;
  L.esrc = q;    q += EQ;
  L.edst = q;    q += EQ;
  L.newid = q;   q += NMAX;
  L.misc = q;
  for (int idx = tid; idx < nn * net.fin0; idx += WGT) L.x[idx] = D.x[(size_t)n0 * net.fin0 + idx];
  for (int e = tid; e < ne; e += WGT) {
    L.esrc[e] = D.esrc[e0 + e];
    L.edst[e] = D.edst[e0 + e];
  }
  const bool norms = pool_norms(L, net, NMAX);
  __syncthreads();
  TP_STAMP(0)
  // ---------------------------------------------------------------- forward levels (tape: rows of the kept nodes)
  // per-level tables in LDS: indexed by the (run-time) level, as private arrays they lived in scratch memory (672 B per
  // lane in round 3)
  __shared__ TapeOffsets off[6];
  __shared__ TapeLevel tps[6];
  __shared__ int nl[6], kl[6], El[6];
  if (tid == 0) {
    tape_layout(net, NMAX, EMAX, off);
    for (int l = 0; l < net.nlevels; ++l) tps[l] = tape_level(ws, off[l]);
  }
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
    if (tid == 0) {
      nl[l] = n;
      El[l] = E;
    }
    run_level<true>(L, lv, C, net.ratio, n, E, nullptr, rmax, rmean, NMAX, &tps[l],
                    norms ? reinterpret_cast<const float*>(L.misc)[1 + l] : -1.f);
    if (tid == 0) kl[l] = n;
    TP_STAMP(1 + l)
  }
  // ---------------------------------------------------------------- head forward, loss term, head backward
  // (buffers in the conv-output area, which is free from here on)
  const int OUTP = (OUT + 3) & ~3;
  float* a0 = L.h;              // [2C]   embedding
  float* a1 = a0 + 2 * C;       // [128]  relu(lin1)
  float* a2 = a1 + 128;         // [64]   relu(lin2)
  float* pr = a2 + 64;          // [OUT]  head outputs
  float* dz3 = pr + OUTP;       // [OUT]
  float* dz2 = dz3 + OUTP;      // [64]
  float* dz1 = dz2 + 64;        // [128]
  float* scratch = dz1 + 128;   // [WGT]
  float* scal = scratch + WGT;  // [4]: d loss / d q, index of q
  if (tid < C) {
    a0[tid] = rmax;
    a0[C + tid] = rmean;
  }
  __syncthreads();
  head_fwd(a0, 2 * C, net.lin1_w, net.lin1_b, 128, a1, scratch);
  if (tid < 128) a1[tid] = fmaxf(a1[tid], 0.f);
  __syncthreads();
  head_fwd(a1, 128, net.lin2_w, net.lin2_b, 64, a2, scratch);
  if (tid < 64) a2[tid] = fmaxf(a2[tid], 0.f);
  __syncthreads();
  head_fwd(a2, 64, net.lin3_w, net.lin3_b, OUT, pr, scratch);
  TP_STAMP(8)
  if (tid < 64) {   // wave 0: softmax, q, the loss term and d loss / d q
    const int lane = tid;
    if (net.softmax) {
      float mx = -INFINITY;
      for (int c = lane; c < OUT; c += 64) mx = fmaxf(mx, pr[c]);
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float s = 0.f;
      for (int c = lane; c < OUT; c += 64) s += expf(pr[c] - mx);
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      for (int c = lane; c < OUT; c += 64) pr[c] = expf(pr[c] - mx) / s;
    }
    // the other network's value (no gradient): max over its outputs (mode 0) or its output of the action (mode 1)
    const float* qo = D.q_other + (size_t)b * OUT;
    const int act = (int)D.action[b];
    float omx = -INFINITY;
    for (int c = lane; c < OUT; c += 64) omx = fmaxf(omx, qo[c]);
    for (int o = 32; o > 0; o >>= 1) omx = fmaxf(omx, __shfl_xor(omx, o, 64));
    // this network's q: its output of the action (mode 0) or its first maximum (mode 1)
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int c = lane; c < OUT; c += 64) {
      const float v = pr[c];
      if (v > bv) {
        bv = v;
        bi = c;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bv, o, 64);
      const int i2 = __shfl_xor(bi, o, 64);
      if (v2 > bv || (v2 == bv && i2 < bi)) {
        bv = v2;
        bi = i2;
      }
    }
    if (lane == 0) {
      const float nf = D.nonfinal[b], r = D.reward[b], gam = (float)D.gamma;
      float a, t, qv;
      int jq;
      if (D.mode == 0) {   // loss(q1(s)[a], r + gamma max q2(s'))
        a = 1.f;
        t = omx * nf * gam + r;
        jq = act;
        qv = pr[act];
      } else {             // loss(q1(s)[a], r + gamma max q2(s')) with the gradient through q2 (the reference's toggle)
        a = -(nf * gam);
        t = r - qo[act];
        jq = bi;
        qv = bv;
      }
      const float diff = fmaf(a, qv, -t);
      const float ad = fabsf(diff);
      ws[0] = ad < 1.f ? 0.5f * diff * diff : ad - 0.5f;                 // HuberLoss(delta = 1)
      scal[0] = a * fminf(fmaxf(diff, -1.f), 1.f) / (float)D.B;          // d (mean loss) / d q
      scal[1] = __int_as_float(jq);
    }
  }
  __syncthreads();
  if (D.out)
    for (int c = tid; c < OUT; c += WGT) D.out[(size_t)b * OUT + c] = pr[c];
  {
    const float dq = scal[0];
    const int jq = __float_as_int(scal[1]);
    if (tid < OUT) {
      if (net.softmax)
        dz3[tid] = pr[tid] * ((tid == jq ? 1.f : 0.f) - pr[jq]) * dq;
      else
        dz3[tid] = tid == jq ? dq : 0.f;
    }
  }
  __syncthreads();
  const mdq_gcn_grad_layout& G = D.layout;
  TP_STAMP(9)
  head_bwd_w(dz3, OUT, a2, 64, gp + G.lin3_w, gp + G.lin3_b);
  head_bwd_in(dz3, OUT, net.lin3_w, 64, dz2);
  if (tid < 64) dz2[tid] = a2[tid] > 0.f ? dz2[tid] : 0.f;
  __syncthreads();
  head_bwd_w(dz2, 64, a1, 128, gp + G.lin2_w, gp + G.lin2_b);
  head_bwd_in(dz2, 64, net.lin2_w, 128, dz1);
  if (tid < 128) dz1[tid] = a1[tid] > 0.f ? dz1[tid] : 0.f;
  __syncthreads();
  head_bwd_w(dz1, 128, a0, 2 * C, gp + G.lin1_w, gp + G.lin1_b);
  head_bwd_in(dz1, 128, net.lin1_w, 2 * C, gemb);
  TP_STAMP(10)
  // ---------------------------------------------------------------- levels, backwards
  float* dxo = L.x;               // [k][C]   d loss / d pooled output of the level; later d loss / d its input
  float* dpre = L.agg;            // [k][C]   d loss / d conv output (before relu) of the kept rows
  float* dagg = L.h;              // [k][fin] d loss / d aggregated input rows
  float* droot = L.h + KM * C;    // [k][fin] d loss / d own input rows (SAGE root weight)
  const int last = net.nlevels - 1;
  for (int l = last; l >= 0; --l) {
    const int nin = nl[l], k = kl[l], El_ = El[l], fin = net.levels[l].fin, type = net.levels[l].type;
    const float* wl = net.levels[l].w_l;
    const float* wr = net.levels[l].w_r;
    const TapeLevel& T = tps[l];
    // ---- readout gradients onto what the next level passed down
    for (int idx = tid; idx < k * C; idx += WGT) {
      const int r = idx / C, c = idx - r * C;
      float g = l == last ? 0.f : dxo[idx];
      g += gemb[C + c] / (float)k;
      if (T.amax[c] == r) g += gemb[c];
      dxo[idx] = g;
    }
    for (int c = tid; c < C; c += WGT) L.deg[c] = net.levels[l].pool_w[c];
    __syncthreads();
    float wn = 0.f;
#pragma unroll 16
    for (int c = 0; c < C; ++c) wn = fmaf(L.deg[c], L.deg[c], wn);
    wn = sqrtf(wn);
    // ---- TopKPooling: d score = d x' . h, through the tanh; 16 lanes per kept row
    for (int r0 = 0; r0 < k; r0 += WGT / 16) {
      const int r = r0 + tid / 16, sub = tid % 16;
      float ds = 0.f, rw = 0.f;
      if (r < k)
        for (int c = sub; c < C; c += 16) {
          const float hv = T.hsel[r * C + c];
          ds = fmaf(dxo[r * C + c], hv, ds);
          rw = fmaf(hv, L.deg[c], rw);
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        ds += __shfl_xor(ds, o, 16);
        rw += __shfl_xor(rw, o, 16);
      }
      if (r < k && sub == 0) {
        const float s = T.ssel[r];
        L.score[r] = ds * (1.f - s * s);   // d loss / d (h . w / |w|)
        rawb[r] = rw / wn;
      }
    }
    __syncthreads();
    // ---- d conv output of the kept rows (relu mask), TopKPooling weight gradient
    for (int idx = tid; idx < k * C; idx += WGT) {
      const int r = idx / C, c = idx - r * C;
      const float hv = T.hsel[idx];
      dpre[idx] = hv > 0.f ? fmaf(dxo[idx], T.ssel[r], L.score[r] * L.deg[c] / wn) : 0.f;
    }
    if (tid < C) {
      float g = 0.f;
      const float wc = L.deg[tid];
      for (int r = 0; r < k; ++r) g = fmaf(L.score[r], T.hsel[r * C + tid] / wn - rawb[r] * wc / (wn * wn), g);
      gp[G.pool_w[l] + tid] = g;
    }
    __syncthreads();
    // ---- weight gradients: (kept rows of d pre)^T (kept rows of agg | x), torch layout [C][fin]
    for (int idx = tid; idx < C * fin; idx += WGT) {
      const int c = idx / fin, f = idx - c * fin;
      float gl = 0.f, gr = 0.f;
      for (int r = 0; r < k; ++r) {
        const float d = dpre[r * C + c];
        gl = fmaf(d, T.aggsel[r * fin + f], gl);
        if (type == 0) gr = fmaf(d, T.xsel[r * fin + f], gr);
      }
      gp[G.w_l[l] + idx] = gl;
      if (type == 0) gp[G.w_r[l] + idx] = gr;
    }
    if (tid < C) {
      float g = 0.f;
      for (int r = 0; r < k; ++r) g += dpre[r * C + tid];
      gp[G.b[l] + tid] = g;
    }
    TP_STAMP(11 + 2 * l)
    if (l == 0) break;   // nothing below the input features
    // ---- input gradient: d agg = d pre W_l, d own row = d pre W_r (one wave per (row, feature), lanes over c)
    {
      const int lane = tid & 63, wave = tid >> 6;
      for (int pair = wave; pair < k * fin; pair += WGT / 64) {
        const int r = pair / fin, f = pair - r * fin;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
          const float d = dpre[r * C + c];
          s1 = fmaf(d, wl[(size_t)f * C + c], s1);
          if (type == 0) s2 = fmaf(d, wr[(size_t)f * C + c], s2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (lane == 0) {
          dagg[pair] = s1;
          droot[pair] = s2;
        }
      }
    }
    // the level's graph: edges from the tape, rank of a kept node, in-degrees
    for (int e = tid; e < El_; e += WGT) {
      L.esrc[e] = T.esrc[e];
      L.edst[e] = T.edst[e];
    }
    for (int i = tid; i < nin; i += WGT) L.newid[i] = -1;
    __syncthreads();
    for (int r = tid; r < k; r += WGT) L.newid[T.perm[r]] = r;
    for (int i = tid; i < nin; i += WGT) {
      int cnt = 0;
      for (int e = 0; e < El_; ++e) cnt += L.edst[e] == i;
      L.score[i] = type == 0 ? (float)(cnt > 0 ? cnt : 1) : 1.0f / sqrtf((float)(cnt + 1));
    }
    __syncthreads();
    // segmented scatter-add along the edges, in edge order, one thread per feature (deterministic sums)
    if (tid < fin) {
      const int f = tid;
      for (int i = 0; i < nin; ++i) dxo[i * fin + f] = 0.f;
      if (type == 0) {
        for (int r = 0; r < k; ++r) dxo[T.perm[r] * fin + f] += droot[r * fin + f];
        for (int e = 0; e < El_; ++e) {
          const int d = L.edst[e], r = L.newid[d];
          if (r >= 0) dxo[L.esrc[e] * fin + f] += dagg[r * fin + f] / L.score[d];
        }
      } else {
        for (int r = 0; r < k; ++r) {
          const int i = T.perm[r];
          dxo[i * fin + f] += L.score[i] * L.score[i] * dagg[r * fin + f];
        }
        for (int e = 0; e < El_; ++e) {
          const int d = L.edst[e], r = L.newid[d], s = L.esrc[e];
          if (r >= 0) dxo[s * fin + f] += L.score[s] * L.score[d] * dagg[r * fin + f];
        }
      }
    }
    __syncthreads();
    TP_STAMP(12 + 2 * l)
  }
}

// flat gradient = sum over the graphs (in graph order) of their slices; loss = mean of the loss terms
__global__ __launch_bounds__(256) void gcn_grad_reduce_kernel(int B, int total, const float* partial, float* grad,
                                                               const float* workspace, int wstride, float* loss) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += partial[(size_t)b * total + i];
    grad[i] = s;
  }
  if (i == 0 && loss) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += workspace[(size_t)b * wstride];
    *loss = s / (float)B;
  }
}

// dst (kernels' layout) <- src (torch layout): per segment a [rows][cols] matrix written transposed ([cols][rows]),
// or a plain copy (cols == 1)
__global__ __launch_bounds__(256) void gcn_pack_kernel(mdq_gcn_pack_table T) {
  const int s = blockIdx.y;
  if (s >= T.n) return;
  const int rows = T.rows[s], cols = T.cols[s];
  const float* src = T.src[s];
  float* dst = T.dst[s];
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * cols; idx += gridDim.x * 256) {
    const int c = idx / rows, r = idx - c * rows;       // consecutive threads write consecutive dst entries
    dst[idx] = src[(size_t)r * cols + c];
  }
}

}  // namespace mdq_gcn

extern "C" int64_t mdq_gcn_train_workspace(const mdq_gcn_net* net, int32_t NMAX, int32_t EMAX) {
  if (!net || NMAX <= 0 || EMAX <= 0) return -1;
  return mdq_gcn::tape_layout(*net, NMAX, EMAX, nullptr);
}

extern "C" int mdq_gcn_pack(const mdq_gcn_pack_table* table, void* stream) {
  if (!table || table->n <= 0 || table->n > MDQ_GCN_PACK_MAX) return mdq_set_error("mdq_gcn_pack: bad table");
  hipLaunchKernelGGL(mdq_gcn::gcn_pack_kernel, dim3(16, table->n), dim3(256), 0, (hipStream_t)stream, *table);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("gcn_pack_kernel launch failed");
  return 0;
}

extern "C" int mdq_gcn_train_step(const mdq_gcn_net* net, const mdq_gcn_train_desc* d, void* stream) {
  using namespace mdq_gcn;
  if (!net || !d || d->B <= 0 || !d->x || !d->node_ptr || !d->edge_ptr || !d->q_other || !d->action || !d->reward ||
      !d->nonfinal || !d->workspace || !d->partial || !d->grad || !d->loss)
    return mdq_set_error("mdq_gcn_train_step: bad arguments");
  const int C = net->C, NMAX = d->NMAX, EMAX = d->EMAX;
  if (C != 64 && C != 128 && C != 256 && C != 32) return mdq_set_error("mdq_gcn_train_step: conv width must divide 256");
  if (net->out_dim > 256) return mdq_set_error("mdq_gcn_train_step: more than 256 head outputs");
  if (net->fin0 > 32 && NMAX > NACC * (WGT / C)) return mdq_set_error("mdq_gcn_train_step: graph too large for the input width");
  for (int l = 0; l < net->nlevels; ++l) {
    double n = NMAX;
    for (int j = 0; j < l; ++j) n = std::ceil(net->ratio * n);
    if (net->levels[l].fin > 32 && n > NACC * (WGT / C)) return mdq_set_error("mdq_gcn_train_step: too many nodes at a wide level");
    if (l > 0 && net->levels[l].fin != C) return mdq_set_error("mdq_gcn_train_step: inner levels must have conv-width inputs");
  }
  const int KM = (int)std::ceil(net->ratio * (double)NMAX);
  const int OUTP = (net->out_dim + 3) & ~3;
  // the backward buffers live in the conv-output area of the forward pass
  if ((size_t)NMAX * (C + 1) < (size_t)2 * KM * C || (size_t)NMAX * (C + 1) < (size_t)(2 * C + 128 + 64 + 2 * OUTP + 64 + 128 + WGT + 4))
    return mdq_set_error("mdq_gcn_train_step: NMAX too small for the backward buffers");
  if (mdq_gcn_xs(*net, NMAX) < KM * C) return mdq_set_error("mdq_gcn_train_step: level buffer too small");
  size_t lds = sizeof(float) * ((size_t)mdq_gcn_xs(*net, NMAX) * 2 + (size_t)NMAX * (C + 1) + (size_t)NMAX +
                                (size_t)(NMAX > C ? NMAX : C) + 2 * (size_t)C + ((KM + 3) & ~3)) +
               sizeof(int) * ((size_t)NMAX + 4 + 3 * ((size_t)EMAX + 3) + NMAX + 8 + WGT / 64) + 16;
  if (lds > 160 * 1024) return mdq_set_error("mdq_gcn_train_step: graph does not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gcn_train_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  const int wstride = tape_layout(*net, NMAX, EMAX, nullptr);
  hipLaunchKernelGGL(gcn_train_kernel, dim3(d->B), dim3(WGT), lds, st, *net, *d, wstride);
  e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  hipLaunchKernelGGL(gcn_grad_reduce_kernel, dim3((d->layout.total + 255) / 256), dim3(256), 0, st, d->B, d->layout.total,
                     d->partial, d->grad, d->workspace, wstride, d->loss);
  e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  return 0;
}
