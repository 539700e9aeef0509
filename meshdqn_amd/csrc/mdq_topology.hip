// Device version of the batched topology engine (mdq_env_topology_host): one 512-thread workgroup per environment
// derives, from coordinates + cells alone and entirely in LDS, everything the environment step needs from a mesh:
//   unique edges numbered by first appearance / P2 dof map / dof coordinates (MeshTopology, flow_solver.py:85-86),
//   boundary vertices + airfoil facets (flow_solver.py:194-226), `removable` (numpy-`in` quirk, flow_solver.py:75-78),
//   polygon distances + stable argsort + N-closest window (Env2DAirfoil.py:220-241, 293-315), the state graph
//   (Env2DAirfoil.py:258-280) and, optionally, the index data of the matrix-free IPCS path (mdq_ipcs_topo_out).
// Sequential semantics (first-appearance edge ids, ordered lists, stable sort) are reproduced with parallel
// primitives: an LDS hash table with atomicMin on the slot index, flag -> exclusive scan -> scatter compactions and
// rank-by-counting sorts.  Floating point follows the host engine operation by operation (no FMA contraction, IEEE
// sqrt / division), so every output array is bit-identical to mdq_env_topology_host's.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_topo {

#ifndef MDQ_TOPO_TW
#define MDQ_TOPO_TW 1024
#endif
constexpr int TW = MDQ_TOPO_TW;
// Capacities as a template parameter K (round 4), as in mdq_remesh.hip: K = 1 - 1024 vertices / 2048 triangles / 3072
// edges, every table in LDS (ids fit 10 / 12 bits); K = 4 - 4096 / 8192 / 12288 (the red-refined lab meshes of BASELINE
// configs[4]), the same code with every table on a slab in GLOBAL memory per mesh.  All byte offsets of the table layout
// scale with K; the index data of the matrix-free IPCS path (packed 12-bit dof ids) exists for K = 1 only.
template <int K>
struct TCap {
  static constexpr int NV = 1024 * K, NT = 2048 * K, NE = 3072 * K, NP = NV + NE, NS = 3 * NT, HS = 8192 * K;
  static constexpr int VBITS = K == 1 ? 10 : (K == 4 ? 12 : 14), HSHIFT = K == 1 ? 19 : (K == 4 ? 17 : 15), PER = NS / TW;
  // (the owner SLOT of an edge - 3 * cell + local edge - is beyond 16 bits from 21 846 triangles on: K = 16)
  using own_t = typename std::conditional<(K > 4), uint32_t, uint16_t>::type;
  static constexpr size_t BASE = ((size_t)16 * NV + 8 * HS + sizeof(int) * NS + sizeof(uint16_t) * NS + 2 * sizeof(uint16_t) * NE + sizeof(own_t) * NE +
                                  NE + NV + sizeof(int) * (TW + 16) + 15) & ~(size_t)15;
  // (K = 16: + the staging array of the dof <- slot lists - slot ids 6 t + i are beyond 16 bits there, and 32-bit entries do not
  //  fit in front of `fill` in region R)
  using stage_t = typename std::conditional<(K > 4), uint32_t, uint16_t>::type;
  static constexpr size_t BYTES = BASE + (K > 4 ? sizeof(uint32_t) * 6 * NT : 0);
};
// PER flags of a thread as bits (a `bool[PER]` is PER registers: 96 of them in the 16 384-vertex instance)
template <int PER>
struct Flags {
  uint32_t w[(PER + 31) / 32];
  __device__ __forceinline__ void set(int i, bool v) {
    if ((i & 31) == 0) w[i >> 5] = 0u;
    w[i >> 5] |= v ? 1u << (i & 31) : 0u;
  }
  __device__ __forceinline__ bool get(int i) const { return (w[i >> 5] >> (i & 31)) & 1u; }
};
constexpr int TNV = TCap<1>::NV, TNT = TCap<1>::NT, TNP = TCap<1>::NP;     // (the K = 1 capacities: host-side checks)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int TNPOLY = 256;            // polygon points of the K = 1 / 4 instances (K = 16: 512, `TPOLY` in the kernel)

// exclusive scan of a[0..n) in place (n <= PER * TW), returns the total; `part` = TW ints of scratch
template <int PER>
__device__ __forceinline__ int scan_excl(int* a, int n, int* part) {
  const int tid = threadIdx.x;
  // (the running sums of the thread's PER entries are NOT kept across the barriers below - PER registers, spilled by the 1 024-thread
  //  large-mesh instance at its 128-VGPR cap - but formed again from the entries in the write-back pass)
  int run = 0, tid1 = tid;
  asm volatile("" : "+v"(tid1));       // (per call: shared between the ten calls of a launch, the addresses were spilled at kernel start)
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = tid1 * PER + i;
    run += idx < n ? a[idx] : 0;
  }
  // thread totals: inclusive scan inside the wave by shuffles, then the TW / 64 wave totals through LDS
  // (two barriers; the Hillis-Steele scan over the 512 thread totals took 19, ten times per launch)
  const int lane = tid & 63, wave = tid >> 6;
  int v = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  __syncthreads();   // (the previous call's readers of `part` are done)
  if (lane == 63) part[wave] = v;
  __syncthreads();
  int pre = 0, total = 0;
#pragma unroll
  for (int w = 0; w < TW / 64; ++w) {
    const int wt = part[w];
    if (w < wave) pre += wt;
    total += wt;
  }
  int at = pre + v - run;
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));       // (the entries' addresses - 64-bit on the slab - are formed again, not kept across the barriers)
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = tid2 * PER + i;
    if (idx < n) {
      const int val = a[idx];
      a[idx] = at;
      at += val;
    }
  }
  __syncthreads();
  return total;
}

template <int SH>
__device__ __forceinline__ uint32_t hslot_t(uint32_t key) { return (key * 2654435761u) >> SH; }

// squared point-segment distance, operation order of the host engine (seg_dist2)
__device__ __forceinline__ double seg_dist2(double px, double py, double ax, double ay, double bx, double by) {
#pragma clang fp contract(off)
  const double abx = bx - ax, aby = by - ay;
  double t = ((px - ax) * abx + (py - ay) * aby) / (abx * abx + aby * aby);
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  const double qx = ax + t * abx - px, qy = ay + t * aby - py;
  return qx * qx + qy * qy;
}

#ifdef MDQ_TOPO_TRACE
// debug build only: s_memtime deltas of thread 0 of mesh 0 at the section boundaries
__device__ long long mdq_topo_trace_buf[16];
#define TT_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_topo_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
extern "C" int mdq_topo_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_topo_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_topo_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
// (K = 4: the lane index is made opaque at every section boundary - the per-lane 64-bit table addresses of the slab
//  instance, hoisted across sections, were what the register allocator spilled)
#define TT_STAMP(k) { if (K != 1) asm volatile("" : "+v"(tid)); }
#endif

template <int K>
__global__ __launch_bounds__(TW) void topology_kernel(mdq_env_topo_desc D, mdq_ipcs_topo_out O, int has_ipcs,
                                                      int32_t* status, unsigned char* slab, mdq_topo_handover HO) {
#pragma clang fp contract(off)
#ifdef MDQ_TOPO_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  using CP = TCap<K>;
  constexpr int TNV = CP::NV, TNE = CP::NE, TNP = CP::NP, TNS = CP::NS, HSZ = CP::HS, PER = CP::PER;
  constexpr int TNT = CP::NT;
  (void)TNT;
  // polygon points: 256 (the airfoil of the lab meshes and of their first refinement); the 16 384-vertex instance 512 (the
  // twice-refined ys930 carries ~480): segment ids of the candidate queue in 16 bits, group masks of 128 bits, 128 groups
  constexpr int TPOLY = K > 4 ? 512 : TNPOLY, NGMAX = TPOLY / 4;
  using cq_t = typename std::conditional<(K > 4), unsigned long long, uint32_t>::type;
  using fm_t = typename std::conditional<(K > 4), unsigned __int128, unsigned long long>::type;
  constexpr int CQB = K > 4 ? 16 : 8;
  // K = 4 (round 5): the EDGE HASH lives in LDS - 16 384 slots x (key, value) = 128 KB, load <= 0.75 at the capacity of 12 288
  // edges - and, once the edge phase is over, the same LDS holds the vertex COORDINATES (64 KB): the hash inserts / probes
  // (atomics and dependent reads on the slab: an L2 round trip each) were 47 % of the large-mesh instance and the coordinate
  // reads most of the polygon-distance section's 27 % (tools/trace_topo.py on the refined ys930).  Every other table stays on
  // the slab; the numbering of the edges does not depend on the table (first appearance by slot index).
  constexpr int HT = K == 4 ? 16384 : HSZ;                   // hash slots (K = 16: 131 072, on the slab like every other table)
  auto hslot = [](uint32_t key) { return hslot_t<(K == 4 ? 18 : CP::HSHIFT)>(key); };
  // (always inlined: with a seventh call site the compiler kept it as a FUNCTION - a call inside the kernel, 288 B of stack
  //  per lane and every section slower)
  auto scan_excl = [](int* a_, int n_, int* part_) __attribute__((always_inline)) { return mdq_topo::scan_excl<CP::PER>(a_, n_, part_); };
  extern __shared__ __align__(16) unsigned char lds_[];
  unsigned char* smem = K == 1 ? lds_ : slab + (size_t)blockIdx.x * ((CP::BYTES + 255) & ~(size_t)255);
  double2* X = reinterpret_cast<double2*>(smem);                              // [TNV]
  uint32_t* hkey = K == 4 ? reinterpret_cast<uint32_t*>(lds_) : reinterpret_cast<uint32_t*>(smem + 16 * TNV);   // [HT] | region R (64 KB
  uint32_t* hval = hkey + HT;                                                 // [HT]   | for K = 1), re-used after the edge phase
  int* scanb = reinterpret_cast<int*>(smem + 16 * TNV + 8 * HSZ);             // [TNS]
  uint16_t* eid_slot = reinterpret_cast<uint16_t*>(scanb + TNS);              // [TNS]
  uint16_t* ea = eid_slot + TNS;                                              // [TNE]
  uint16_t* eb = ea + TNE;
  using own_t = typename CP::own_t;
  own_t* eown = reinterpret_cast<own_t*>(eb + TNE);                           // [TNE] owner slot of the edge
  uint8_t* eflag = reinterpret_cast<uint8_t*>(eown + TNE);                    // [TNE] bit0 boundary, bits 4-6 tag + 1
  uint8_t* onb = eflag + TNE;                                                 // [TNV]
  int* part = reinterpret_cast<int*>(onb + TNV);                              // [TW]
  int* misc = part + TW;                                                      // [16]
  // region R after the edge phase
  unsigned char* R = smem + 16 * TNV;
  constexpr int RPOLY = 16 * TNV, RSEG = RPOLY + 16 * TPOLY;                 // (K = 1: 16384 / 20480)
  uint16_t* blist = reinterpret_cast<uint16_t*>(R);                           // [TNV] boundary vertices
  uint16_t* remv = blist + TNV;                                               // [TNV] removable vertices (ascending)
  uint16_t* order = remv + TNV;                                               // [TNV]
  int16_t* inv = reinterpret_cast<int16_t*>(order + TNV);                     // [TNV]
  // (K = 4: the distances and the histograms of the two counting sorts in the upper half of the LDS too, behind the polygon tables:
  //  lds_ + 64 KB: polygon tables 20 KB | histograms 8 KB | distances 32 KB)
  double* dist = K == 4 ? reinterpret_cast<double*>(lds_ + 65536 + 28672) : reinterpret_cast<double*>(R + 8 * TNV);   // [TNV]  (K = 16: on the slab)
  // (K = 4: the polygon and the small tables of the distance section - 16 KB that every vertex gathers from - sit in the upper
  //  half of the LDS, free once the edge hash is dead; the coordinates take the lower half)
  // (K = 16: the lower half of the LDS is free - hash and coordinates are on the slab -: the polygon tables, twice as large, there)
  unsigned char* PB = K == 1 ? R + RPOLY : (K == 4 ? lds_ + 65536 : lds_);    // poly | segf | segr | ye | pmx
  unsigned char* GB = K == 1 ? nullptr : (K == 4 ? lds_ + 65536 + 16384 : lds_ + 32768);   // grp | yrange | chf | chr   (K = 1: in R2, below)
  double2* poly = reinterpret_cast<double2*>(PB);                             // [TPOLY]
  int* cntd = reinterpret_cast<int*>(R + RSEG);                               // [TNP + 1] counts / pointers (IPCS phase)
  int* fill = cntd + TNP + 8;                                                 // [TNP]

  const int b = blockIdx.x;
  int tid = threadIdx.x;
  const int nv = D.nv[b], nt = D.nt[b];
  const int64_t Bq = b;
  const double* xg = D.coords + Bq * D.NV * 2;
  const int32_t* tri = D.cells + Bq * D.NT * 3;
  int32_t* cd = D.cell_dofs + Bq * 6 * D.NT;
  double* pts = D.points + Bq * D.NP * 2;
  if (tid == 0) status[b] = 0;
  // (HO: the mesh and - below - its edge numbering also go to another engine's input arrays, see mdq_topo_handover)
  int32_t* hcd = HO.cell_dofs ? HO.cell_dofs + Bq * 6 * D.NT : nullptr;
  for (int v = tid; v < TNV; v += TW) {
    if (v < nv) {
      X[v] = make_double2(xg[2 * v], xg[2 * v + 1]);
      if (HO.coords) reinterpret_cast<double2*>(HO.coords)[Bq * D.NV + v] = X[v];
    }
    onb[v] = 0;
  }
  if (HO.cells)
    for (int i = tid; i < 3 * nt; i += TW) HO.cells[Bq * D.NT * 3 + i] = tri[i];
  if (tid == 0) {
    if (HO.nv) HO.nv[b] = nv;
    if (HO.nt) HO.nt[b] = nt;
  }
  for (int h = tid; h < HT; h += TW) {
    hkey[h] = EMPTY;
    hval[h] = 0x7FFFFFFFu;
  }
  __syncthreads();

  TT_STAMP(0)
  // ================= edges numbered by first appearance in (cell, local edge k opposite vertex k) order
  auto slot_key = [&](int s, int& a, int& c) {
    const int t = s / 3, k = s - 3 * t;
    const int va = tri[3 * t + (k == 0 ? 1 : 0)], vb = tri[3 * t + (k == 2 ? 1 : 2)];
    a = min(va, vb);
    c = max(va, vb);
    return ((uint32_t)a << CP::VBITS) | (uint32_t)c;
  };
  auto probe = [&](uint32_t key) {
    uint32_t h = hslot(key);
    // (bounded: a key that was refused by a full table - below - must not spin here)
    for (int n_ = 0; hkey[h] != key && n_ < HT; ++n_) h = (h + 1) & (HT - 1);
    return h;
  };
  const int nslots = 3 * nt;
  const bool edges_given = has_ipcs && O.flow_only && O.cell_dofs_in && O.ne_in;
  int ne_given = 0;
  if (edges_given) {
    // The edge numbering of another engine's run on the same mesh (its cell dofs 3..5 = nv + edge id): owner = lowest slot
    // of an edge, boundary = a single owner, both by LDS atomics on the edge id - the hash insert + three probes per slot
    // of the general path (33 k cycles) are not needed.  Same ea / eb / eown / eflag / eid_slot / cell dofs as that path.
    const int32_t* cin = O.cell_dofs_in + Bq * 6 * D.NT;
    ne_given = O.ne_in[b];
    int* own = reinterpret_cast<int*>(hkey);               // [ne] lowest slot   (the hash arrays are not used on this path)
    int* cnt_ = reinterpret_cast<int*>(hval);              // [ne] owners
    if (ne_given < 0 || ne_given > TNE || nv + ne_given > D.NP) {
      if (tid == 0) status[b] = -1;
      return;
    }
    for (int e = tid; e < ne_given; e += TW) {
      own[e] = 0x7FFFFFFF;
      cnt_[e] = 0;
    }
    __syncthreads();
    for (int s = tid; s < nslots; s += TW) {
      const int t = s / 3, k = s - 3 * t;
      const int e = cin[(3 + k) * D.NT + t] - nv;
      if (e < 0 || e >= ne_given) {
        status[b] = -1;                                    // (not the dofs of this mesh)
        continue;
      }
      eid_slot[s] = (uint16_t)e;
      atomicMin(&own[e], s);
      atomicAdd(&cnt_[e], 1);
      cd[(3 + k) * D.NT + t] = nv + e;
      cd[k * D.NT + t] = tri[3 * t + k];
      if (hcd) {
        hcd[(3 + k) * D.NT + t] = nv + e;
        hcd[k * D.NT + t] = tri[3 * t + k];
      }
    }
    __syncthreads();
    for (int e = tid; e < ne_given; e += TW) {
      int a, c;
      const int s = own[e];
      slot_key(s < nslots ? s : 0, a, c);
      ea[e] = (uint16_t)a;
      eb[e] = (uint16_t)c;
      eown[e] = (own_t)s;
      eflag[e] = cnt_[e] >= 2 ? 0 : 1;
    }
    if (tid == 0) {
      D.ne[b] = ne_given;
      if (HO.ne) HO.ne[b] = ne_given;
    }
    for (int i = tid; i < nv; i += TW) {
      pts[2 * i] = X[i].x;
      pts[2 * i + 1] = X[i].y;
    }
    __syncthreads();
    if (status[b] != 0) return;
  }
  int ne = ne_given;
  if (!edges_given) {
    for (int s = tid; s < nslots; s += TW) {
      int a, c;
      const uint32_t key = slot_key(s, a, c);
      uint32_t h = hslot(key);
      int n_ = 0;
      for (;; ++n_) {
        const uint32_t old = atomicCAS(&hkey[h], EMPTY, key);
        if (old == EMPTY || old == key || n_ >= HT) break;
        h = (h + 1) & (HT - 1);
      }
      if (n_ >= HT) {
        status[b] = -1;                                    // more distinct edges than hash slots: not a mesh within the capacities
        continue;
      }
      atomicMin(&hval[h], (uint32_t)s);
    }
    __syncthreads();
    if (K != 1 && status[b] != 0) return;                  // (workgroup-uniform: written before the barrier)
    for (int s = tid; s < TNS; s += TW) {
      int first = 0;
      if (s < nslots) {
        int a, c;
        const uint32_t h = probe(slot_key(s, a, c));
        const uint32_t m = hval[h] & 0x7FFFFFFFu;
        first = m == (uint32_t)s;
        if (!first) atomicOr(&hval[h], 0x80000000u);  // a second owner: interior edge
      }
      scanb[s] = first;
    }
    __syncthreads();
    // (the flags must be read back before the scan overwrites them)
    Flags<PER> isfirst;
  #pragma unroll
    for (int i = 0; i < PER; ++i) isfirst.set(i, scanb[tid * PER + i] != 0);
    __syncthreads();
    ne = scan_excl(scanb, TNS, part);
    if (ne > TNE || nv + ne > D.NP) {
      if (tid == 0) status[b] = -1;
      return;
    }
  #pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = tid * PER + i;
      if (s < nslots && isfirst.get(i)) {
        int a, c;
        const uint32_t h = probe(slot_key(s, a, c));
        const int e = scanb[s];
        const bool shared = (hval[h] & 0x80000000u) != 0;
        ea[e] = (uint16_t)a;
        eb[e] = (uint16_t)c;
        eown[e] = (own_t)s;
        eflag[e] = shared ? 0 : 1;
      }
    }
    __syncthreads();
    for (int s = tid; s < nslots; s += TW) {
      int a, c;
      const uint32_t h = probe(slot_key(s, a, c));
      const int e = scanb[hval[h] & 0x7FFFFFFFu];
      eid_slot[s] = (uint16_t)e;
      const int t = s / 3, k = s - 3 * t;
      cd[(3 + k) * D.NT + t] = nv + e;
      cd[k * D.NT + t] = tri[3 * t + k];
      if (hcd) {
        hcd[(3 + k) * D.NT + t] = nv + e;
        hcd[k * D.NT + t] = tri[3 * t + k];
      }
    }
    if (tid == 0) {
      D.ne[b] = ne;
      if (HO.ne) HO.ne[b] = ne;
    }
    for (int i = tid; i < nv; i += TW) {
      pts[2 * i] = X[i].x;
      pts[2 * i + 1] = X[i].y;
    }
    __syncthreads();
  }
  if (K == 4) {
    // the hash is dead: the coordinates move from the slab into its LDS (every later section gathers them; K = 16: they stay
    // on the slab - 207 KB for the twice-refined ys930)
    double2* Xl = reinterpret_cast<double2*>(lds_);
    for (int v = tid; v < nv; v += TW) Xl[v] = X[v];
    __syncthreads();
    X = Xl;
  }
  const int n2 = nv + ne;
  TT_STAMP(1)
  // ================= P2 dof coordinates, boundary vertices, facet tags (later marks override earlier ones)
  const double E = 3.0e-16;
  double ymin = 1e300, ymax = -1e300;
  for (int i = tid; i < nv; i += TW) {
    ymin = fmin(ymin, X[i].y);
    ymax = fmax(ymax, X[i].y);
  }
  for (int e = tid; e < ne; e += TW) {
    const double2 A = X[ea[e]], Bp = X[eb[e]];
    const double mx = 0.5 * (A.x + Bp.x), my = 0.5 * (A.y + Bp.y);
    pts[2 * (nv + e)] = mx;
    pts[2 * (nv + e) + 1] = my;
    if (eflag[e] & 1) {
      onb[ea[e]] = 1;
      onb[eb[e]] = 1;
      const double Xs[3] = {A.x, Bp.x, mx}, Ys[3] = {A.y, Bp.y, my};
      bool walls = true, air = true, inflow = true, outflow = true;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        walls = walls && (Ys[q] > 0.5 - 2 * E || Ys[q] < -0.5 + 2 * E);
        air = air && (Xs[q] < 3.0 - E && Xs[q] > -0.5 + E && Ys[q] < 0.5 - E && Ys[q] > -0.5 + E);
        inflow = inflow && (Xs[q] < -0.5 + E);
        outflow = outflow && (Xs[q] > 3.0 - 2 * E);
      }
      int tg = 4;
      if (walls) tg = 0;
      if (air) tg = 1;
      if (inflow) tg = 2;
      if (outflow) tg = 3;
      eflag[e] = (uint8_t)(1 | ((tg + 1) << 4));
    }
  }
  // y range of the mesh (inflow profile): workgroup min / max through LDS atomics on the bit patterns is awkward for
  // doubles; 512 partial values through `dist` (free until the distance phase) instead
  __syncthreads();   // (the hash region is dead from here on)
  dist[tid] = ymin;
  dist[TW + tid] = ymax;
  __syncthreads();
  for (int off = TW / 2; off > 0; off >>= 1) {
    if (tid < off) {
      dist[tid] = fmin(dist[tid], dist[tid + off]);
      dist[TW + tid] = fmax(dist[TW + tid], dist[TW + tid + off]);
    }
    __syncthreads();
  }
  const double bot = dist[0], top = dist[TW];
  __syncthreads();
  // airfoil facets in edge order
  for (int e = tid; e < TNS; e += TW) scanb[e] = (e < ne && (eflag[e] >> 4) == 2) ? 1 : 0;
  __syncthreads();
  Flags<PER> flg;
#pragma unroll
  for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
  __syncthreads();
  const int naf = scan_excl(scanb, TNS, part);
  if (naf > D.NAF) {
    if (tid == 0) status[b] = -2;
    return;
  }
  {
    int32_t* af = D.af_facets + Bq * D.NAF * 2;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid * PER + i;
      if (e < ne && flg.get(i)) {
        const int q = scanb[e];
        af[2 * q] = eown[e] / 3;
        af[2 * q + 1] = eown[e] % 3;
      }
    }
  }
  if (tid == 0) D.naf[b] = naf;
  __syncthreads();
  TT_STAMP(2)
  if (!(has_ipcs && O.flow_only)) {   // (an engine that only feeds the IPCS step skips the selection and the state graph)
  // ================= removable: neither x nor y equals ANY boundary vertex's x / y (numpy `in` quirk)
  for (int v = tid; v < TNS; v += TW) scanb[v] = (v < nv && onb[v]) ? 1 : 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
  __syncthreads();
  const int nb = scan_excl(scanb, TNS, part);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = tid * PER + i;
    if (v < nv && flg.get(i)) blist[scanb[v]] = (uint16_t)v;
  }
  __syncthreads();
  // Two hash SETS of the boundary vertices' x and y values (bit patterns; x + 0.0 folds -0 into +0, the one pair of
  // different patterns that compares equal; a NaN equals nothing and is not inserted) at the end of region R, which is free
  // here: a vertex then takes two look-ups instead of comparing with every boundary vertex (876 x ~190 x 2 comparisons,
  // 25 k of the kernel's 175 k cycles).  The same predicate, exactly.  More boundary vertices than half a table: the loop.
  constexpr int TSZ = 1024 * (K > 4 ? 4 : K);     // (K = 16: the LDS tables of K = 4; more boundary vertices than half a table: the loop)
  // (K = 4: in the upper half of the LDS - free between the edge hash and the polygon tables -: the inserts are atomics)
  unsigned long long* hx = K == 1 ? reinterpret_cast<unsigned long long*>(R + 8 * HSZ - 16 * TSZ)
                                  : reinterpret_cast<unsigned long long*>(lds_ + 65536);
  unsigned long long* hy = hx + TSZ;
  const bool hashed = nb <= TSZ / 2;
  auto hpos = [](unsigned long long bits) { return (int)((bits * 0x9E3779B97F4A7C15ull) >> (64 - 10 - (K == 1 ? 0 : 2))); };
  if (hashed) {
    for (int i = tid; i < 2 * TSZ; i += TW) hx[i] = ~0ull;
    __syncthreads();
    for (int j = tid; j < 2 * nb; j += TW) {
      const double2 q = X[blist[j >> 1]];
      const double val = (j & 1) ? q.y : q.x;
      if (val == val) {
        unsigned long long* tab = (j & 1) ? hy : hx;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val + 0.0);
        int h = hpos(bits);
        while (true) {
          const unsigned long long old = atomicCAS(&tab[h], ~0ull, bits);
          if (old == ~0ull || old == bits) break;
          h = (h + 1) & (TSZ - 1);
        }
      }
    }
    __syncthreads();
  }
  auto hfind = [&](const unsigned long long* tab, double val) {
    if (!(val == val)) return 0;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(val + 0.0);
    int h = hpos(bits);
    while (true) {
      const unsigned long long k_ = tab[h];
      if (k_ == bits) return 1;
      if (k_ == ~0ull) return 0;
      h = (h + 1) & (TSZ - 1);
    }
  };
  // (fallback) boundary coordinates packed (the distance buffer is free until the next section): the comparison loop reads
  // consecutive entries, independent of each other, instead of an index and a dependent gather per boundary vertex
  double2* bxy = reinterpret_cast<double2*>(dist);   // [nb] <= TNV / 2 entries fit the TNV doubles of `dist`
  const bool packed = nb <= TNV / 2;
  if (packed && !hashed)
    for (int j = tid; j < nb; j += TW) bxy[j] = X[blist[j]];
  __syncthreads();
  for (int v = tid; v < TNS; v += TW) {
    int r = 0;
    if (v < nv) {
      const double px = X[v].x, py = X[v].y;
      int hit = 0;
      if (hashed) {
        hit = hfind(hx, px) | hfind(hy, py);
      } else if (packed) {
#pragma unroll 8
        for (int j = 0; j < nb; ++j) {
          const double2 q = bxy[j];
          hit |= (q.x == px) | (q.y == py);
        }
      } else {
        for (int j = 0; j < nb; ++j) {
          const double2 q = X[blist[j]];
          hit |= (q.x == px) | (q.y == py);
        }
      }
      r = hit ? 0 : 1;
    }
    scanb[v] = r;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
  __syncthreads();
  const int nrem = scan_excl(scanb, TNS, part);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = tid * PER + i;
    if (v < nv && flg.get(i)) remv[scanb[v]] = (uint16_t)v;
  }
  if (tid == 0) D.nremovable[b] = nrem;
  TT_STAMP(3)
  // ================= distances to the airfoil polygon (0 inside), stable argsort, window of N
  const int np_ = D.npoly;
  for (int i = tid; i < np_; i += TW) poly[i] = make_double2(D.polygon[2 * i], D.polygon[2 * i + 1]);
  __syncthreads();
  // One lane per vertex, segments culled in GROUPS of four, waves made coherent by a counting sort.
  // Exactness is kept the way rounds 2-3 kept it: `seg_dist2` is evaluated on a SUPERSET of the segments that can attain
  // a vertex's minimum distance, the crossing abscissa on exactly the segments its horizontal ray straddles - so minimum
  // and parity are the values of the full (vertex x segment) loop, bit for bit.  What changed is how the superset is found.
  // Rounds 2-3: four lanes per vertex ran an fp32 estimate of EVERY pair (~2 000 instructions per lane, 47 waves: 88 k of the
  // kernel's 175 k cycles, issue-bound).  Now:
  //   (B) per vertex: the bounding circle of every group (<= 64 groups) gives the nearest group (home); fp32 estimates of
  //       the 12 segments of home - 1 .. home + 1 give U = estimate + 2 eta, an upper bound of the true distance + eta
  //       (eta bounds an estimate's absolute error, as before);
  //   (C) the vertices are sorted by (ray can straddle, home): the 64 vertices of a wave sit next to the same few groups;
  //   (D) per vertex, in that order: a group whose circle is farther than U (+ eta) cannot hold the minimum - the group
  //       loop is wave-uniform and a group is skipped when NO lane of the wave needs it (6 - 10 of 30 remain); segments of
  //       the others with an estimate <= U are candidates (bit masks, 32 segments at a time, walked in wave-uniform loops:
  //       the lanes sit at different segments).  A segment attaining the minimum d* has estimate <= d* + eta <= U and lies
  //       in a group whose circle is at most d* away: it is always a candidate.  Straddle bits only for waves that hold a
  //       vertex inside the polygon's y range (no segment straddles the others' rays).
  const int NG = (np_ + 3) >> 2;                                              // groups of 4 segments (<= 64)
  const int NW = (np_ + 31) >> 5;                                             // 32-segment words (<= 8)
  const int NSEG = 32 * NW;                                                   // <= TNPOLY
  float4* segf = reinterpret_cast<float4*>(PB + 16 * TPOLY);                  // [NSEG] {ax, ay, bx - ax, by - ay}   (K = 1: R + RSEG)
  float* segr = reinterpret_cast<float*>(PB + 16 * TPOLY + TPOLY * 16);       // [NSEG] 1 / |b - a|^2
  double* ye = reinterpret_cast<double*>(PB + 16 * TPOLY + TPOLY * 20);       // [NSEG + 1] y of polygon vertex i (closed)
  float* pmx = reinterpret_cast<float*>(ye + TPOLY + 2);                      // [TW / 64] wave maxima of |polygon coordinate|
  unsigned char* R2 = R + 8 * HSZ - 16 * 1024 * K;                            // (the look-up tables of the section above are dead)
  uint16_t* sorder = reinterpret_cast<uint16_t*>(R2 + 4 * TNV);               // [TNV] removable vertices by (band, home)
  uint8_t* keyv = reinterpret_cast<uint8_t*>(R2 + 6 * TNV);                   // [TNV] sort key
  unsigned char* G2 = K == 1 ? R2 + 8 * TNV : GB;
  float4* grp = reinterpret_cast<float4*>(G2);                                // [NGMAX] {cx, cy, radius, -}
  double* yrange = reinterpret_cast<double*>(G2 + 16 * NGMAX);                // [4 * TW / 64] wave minima / maxima of the polygon's y, x
  float4* chf = reinterpret_cast<float4*>(G2 + 16 * NGMAX + 512);             // [NGMAX] group chord {ax, ay, bx - ax, by - ay}
  float2* chr = reinterpret_cast<float2*>(G2 + 16 * NGMAX + 512 + 16 * NGMAX);   // [NGMAX] {1 / |chord|^2, deviation of the group's polyline from it}
  int* hist = K == 1 ? scanb : reinterpret_cast<int*>(lds_ + 65536 + 20480);  // [128]
  {
    float m = 0.f;
    double y0 = 1e300, y1 = -1e300, x0_ = 1e300, x1_ = -1e300;
    for (int i = tid; i <= NSEG; i += TW) {
      if (i < np_) {
        const double2 A = poly[i], Bp = poly[i + 1 == np_ ? 0 : i + 1];
        const double abx = Bp.x - A.x, aby = Bp.y - A.y, l2 = abx * abx + aby * aby;
        segf[i] = make_float4((float)A.x, (float)A.y, (float)abx, (float)aby);
        segr[i] = l2 > 0.0 ? 1.0f / (float)l2 : 0.f;
        ye[i] = A.y;
        y0 = fmin(y0, A.y);
        y1 = fmax(y1, A.y);
        x0_ = fmin(x0_, A.x);
        x1_ = fmax(x1_, A.x);
        m = fmaxf(m, fmaxf(fabsf((float)A.x), fabsf((float)A.y)));
      } else {                                   // padding: far away, never straddled
        if (i < NSEG) {
          segf[i] = make_float4(1.0e18f, 1.0e18f, 0.f, 0.f);
          segr[i] = 0.f;
        }
        ye[i] = poly[0].y;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m = fmaxf(m, __shfl_xor(m, o, 64));
      y0 = fmin(y0, __shfl_xor(y0, o, 64));
      y1 = fmax(y1, __shfl_xor(y1, o, 64));
      x0_ = fmin(x0_, __shfl_xor(x0_, o, 64));
      x1_ = fmax(x1_, __shfl_xor(x1_, o, 64));
    }
    if ((tid & 63) == 0) {
      pmx[tid >> 6] = m;
      yrange[tid >> 6] = y0;
      yrange[TW / 64 + (tid >> 6)] = y1;
      yrange[2 * (TW / 64) + (tid >> 6)] = x0_;
      yrange[3 * (TW / 64) + (tid >> 6)] = x1_;
    }
    for (int i = tid; i < 2 * NGMAX; i += TW) hist[i] = 0;
  }
  __syncthreads();
  float mpoly = 0.f;
  double ylo = 1e300, yhi = -1e300, xlo = 1e300, xhi = -1e300;
#pragma unroll
  for (int w = 0; w < TW / 64; ++w) {
    mpoly = fmaxf(mpoly, pmx[w]);
    ylo = fmin(ylo, yrange[w]);
    yhi = fmax(yhi, yrange[TW / 64 + w]);
    xlo = fmin(xlo, yrange[2 * (TW / 64) + w]);
    xhi = fmax(xhi, yrange[3 * (TW / 64) + w]);
  }
  // (workgroup-uniform values: kept in scalar registers - as five doubles in vector registers they spilled)
  auto uni = [](double v_) {
    const unsigned long long u_ = (unsigned long long)__double_as_longlong(v_);
    const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)u_), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u_ >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));
  };
  ylo = uni(ylo);
  yhi = uni(yhi);
  xlo = uni(xlo);
  xhi = uni(xhi);
  mpoly = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mpoly)));
  const double xmg = 3.0e-14 * fmax(fabs(xlo), fabs(xhi)) + 1e-300;   // (>= the per-segment margins of the crossing test)
  // group circles: centre of the bounding box of the group's <= 5 polygon vertices (fp32), radius = the largest distance
  // of one of them from it, rounded up generously (the group tests below add eta on top)
  for (int g = tid; g < NG; g += TW) {
    float x0 = 3.0e38f, x1 = -3.0e38f, yy0 = 3.0e38f, yy1 = -3.0e38f;
    const int i1 = min(4 * g + 4, np_);                  // last polygon vertex of the group (np_ = vertex 0 again)
    for (int i = 4 * g; i <= i1; ++i) {
      const double2 A = poly[i == np_ ? 0 : i];
      x0 = fminf(x0, (float)A.x);
      x1 = fmaxf(x1, (float)A.x);
      yy0 = fminf(yy0, (float)A.y);
      yy1 = fmaxf(yy1, (float)A.y);
    }
    const float cx = 0.5f * (x0 + x1), cy = 0.5f * (yy0 + yy1);
    float rr = 0.f;
    for (int i = 4 * g; i <= i1; ++i) {
      const double2 A = poly[i == np_ ? 0 : i];
      const float wx = (float)A.x - cx, wy = (float)A.y - cy;
      rr = fmaxf(rr, wx * wx + wy * wy);
    }
    grp[g] = make_float4(cx, cy, sqrtf(rr) * 1.00001f + 1.0e-30f, 0.f);
    // chord first - last vertex of the group and the largest distance of a vertex in between from it (the distance to a
    // segment is convex along the polyline: attained at a vertex): dist(p, group) >= dist(p, chord) - deviation
    const double2 P0 = poly[4 * g], P1 = poly[i1 == np_ ? 0 : i1];
    const float cax = (float)P0.x, cay = (float)P0.y, cbx = (float)(P1.x - P0.x), cby = (float)(P1.y - P0.y);
    const float l2 = cbx * cbx + cby * cby, rl = l2 > 0.f ? 1.0f / l2 : 0.f;
    float dev = 0.f;
    for (int i = 4 * g + 1; i < i1; ++i) {
      const double2 A = poly[i];
      const float wx = (float)A.x - cax, wy = (float)A.y - cay;
      float t = (wx * cbx + wy * cby) * rl;
      t = fminf(fmaxf(t, 0.f), 1.f);
      const float ex = wx - t * cbx, ey = wy - t * cby;
      dev = fmaxf(dev, ex * ex + ey * ey);
    }
    chf[g] = make_float4(cax, cay, cbx, cby);
    chr[g] = make_float2(rl, sqrtf(dev) * 1.001f + 1.0e-30f);
  }
  __syncthreads();
  auto estimate = [&](int i, float pxf, float pyf) {      // fp32 estimate of the squared distance to segment i
    const float4 sg = segf[i];
    const float wx = pxf - sg.x, wy = pyf - sg.y;
    float t = (wx * sg.z + wy * sg.w) * segr[i];
    t = fminf(fmaxf(t, 0.f), 1.f);
    const float cx = wx - t * sg.z, cy = wy - t * sg.w;
    return cx * cx + cy * cy;
  };
  TT_STAMP(6)
  // (B) home group (nearest circle centre: a heuristic, nothing depends on it being the nearest group), sort key
  for (int r = tid; r < nrem; r += TW) {
    const double px = X[remv[r]].x, py = X[remv[r]].y;
    const float pxf = (float)px, pyf = (float)py;
    float best = 3.0e38f;
    int home = 0;
    for (int g = 0; g < NG; ++g) {
      const float4 G = grp[g];
      const float wx = pxf - G.x, wy = pyf - G.y;
      const float dd = wx * wx + wy * wy;
      if (dd < best) {
        best = dd;
        home = g;
      }
    }
    const int key = ((py >= ylo && py < yhi && px >= xlo - xmg && px <= xhi + xmg) ? NGMAX : 0) + home;
    keyv[r] = (uint8_t)key;
    atomicAdd(&hist[key], 1);
  }
  __syncthreads();
  TT_STAMP(7)
  // (C) counting sort by key (any order inside a key: the vertices are independent of each other)
  scan_excl(hist, 2 * NGMAX, part);
  __syncthreads();
  for (int r = tid; r < nrem; r += TW) sorder[atomicAdd(&hist[keyv[r]], 1)] = (uint16_t)r;
  __syncthreads();
  TT_STAMP(8)
  // (D)
  for (int base = 0; base < nrem; base += TW) {
    const int slot = base + tid;
    const bool live = slot < nrem;
    const int r = sorder[min(slot, nrem - 1)];
    const double px = X[remv[r]].x, py = X[remv[r]].y;
    const float pxf = (float)px, pyf = (float)py;
    // absolute error bound of an fp32 distance estimate: <= ~30 ulp of the largest coordinate M (inputs 1 ulp each, the projection parameter 13 M / |ab| + 3.5, the closest point 17 M + 4.5 |ab| + ...); 64 taken
    const float eta = 64.0f * 5.9604645e-8f * fmaxf(mpoly, fmaxf(fabsf(pxf), fabsf(pyf)));
    double d2 = 1e300;
    // candidate queue: up to four segment ids, a byte each (np_ <= 256); a vertex with a fifth candidate (never seen on the
    // lab meshes) is evaluated against every segment instead.  The exact evaluations then take max-over-lanes(candidates)
    // rounds per wave instead of a sum of per-word maxima
    cq_t cq = 0u;
    int cn = 0;
    bool ovf = false;
    auto push = [&](int id, bool c) {
      if (c) {
        if (cn == 4) ovf = true;
        else {
          cq |= (cq_t)id << (CQB * cn);
          ++cn;
        }
      }
    };
    // the 12 segments of the home neighbourhood, per lane (the lanes of a wave read the same few segments: sorted)
    int gi[3];
    float amin = 3.0e38f;
    {
      const int home = keyv[r] & (NGMAX - 1);
#pragma unroll
      for (int dg = 0; dg < 3; ++dg) {
        int g = home + dg - 1;
        g = g < 0 ? NG - 1 : (g >= NG ? 0 : g);
        gi[dg] = g;
#pragma unroll
        for (int k = 0; k < 4; ++k) amin = fminf(amin, estimate(4 * g + k, pxf, pyf));
      }
    }
    const float U = sqrtf(amin) + 2.0f * eta;
    // FOREIGN groups within reach of U (bit g): the home neighbourhood can miss the nearest segment - the circle centres
    // are a heuristic, and near the trailing edge the other surface is as close.  Rare: the group loops below are
    // wave-uniform and skip a group no lane of the wave needs.
    fm_t fm = 0;
    {
      // (the chord test, not the bounding circle: at a distance comparable to a group's size the circle of nearly every
      //  neighbour is within reach - 11 of 12 waves ran the foreign loops, 28 k cycles; 2 eta: the estimate's error twice,
      //  once for the chord, once for the deviation)
      const float Ue = U + 2.0f * eta;
      for (int g = 0; g < NG; ++g) {
        const float4 sg = chf[g];
        const float2 cr = chr[g];
        const float wx = pxf - sg.x, wy = pyf - sg.y;
        float t = (wx * sg.z + wy * sg.w) * cr.x;
        t = fminf(fmaxf(t, 0.f), 1.f);
        const float ex = wx - t * sg.z, ey = wy - t * sg.w;
        const float reach = Ue + cr.y;                                    // (dist(p, chord) - deviation <= U + 2 eta, squared)
        const bool q = live && (ex * ex + ey * ey <= reach * reach * 1.000001f) && g != gi[0] && g != gi[1] && g != gi[2];
        fm |= (fm_t)(q ? 1 : 0) << g;
      }
    }
    const bool foreign = __any(fm != 0);
    if (foreign) {
      for (int g = 0; g < NG; ++g) {
        const bool q = (bool)((fm >> g) & 1);
        if (__any(q)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float a = estimate(4 * g + k, pxf, pyf);
            amin = q ? fminf(amin, a) : amin;
          }
        }
      }
    }
    const float U2 = sqrtf(amin) + 2.0f * eta;              // (<= U: amin only went down)
    const float thr = U2 * U2 * 1.000001f;
    // (the 12 estimates again rather than 12 registers kept across the loops above: the kernel runs 16 waves per CU - 128
    //  registers per lane - and spilled)
#pragma unroll
    for (int dg = 0; dg < 3; ++dg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = 4 * gi[dg] + k;
        push(id, live && estimate(id, pxf, pyf) <= thr);
      }
    }
    if (foreign) {
      const float Ue = U2 + 2.0f * eta;
      for (int g = 0; g < NG; ++g) {
        if (!__any((bool)((fm >> g) & 1))) continue;
        const float4 sg = chf[g];
        const float2 cr = chr[g];
        const float wx = pxf - sg.x, wy = pyf - sg.y;
        float t = (wx * sg.z + wy * sg.w) * cr.x;
        t = fminf(fmaxf(t, 0.f), 1.f);
        const float ex = wx - t * sg.z, ey = wy - t * sg.w;
        const float reach = Ue + cr.y;
        const bool q = (bool)((fm >> g) & 1) && (ex * ex + ey * ey <= reach * reach * 1.000001f);
        if (__any(q)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float a = estimate(4 * g + k, pxf, pyf);
            push(4 * g + k, q && a <= thr);
          }
        }
      }
    }
    {
      // (one copy of the exact evaluation: queue rounds first, then - wave-uniform, rare - every segment for overflowed lanes)
      const bool anyovf = __any(ovf);
      int it = 0;
      while (true) {
        const bool qrun = cn > 0, orun = ovf && it < np_;
        if (!__any(qrun || (anyovf && orun))) break;
        int i = it;
        if (qrun) {
          i = (int)(cq & (cq_t)((1u << CQB) - 1u));
          cq >>= CQB;
          --cn;
        } else {
          ++it;
        }
        if (qrun || orun) {
          const double2 A = poly[i], Bp = poly[i + 1 == np_ ? 0 : i + 1];
          d2 = fmin(d2, seg_dist2(px, py, A.x, A.y, Bp.x, Bp.y));
        }
      }
    }
    // inside test (crossing number of the horizontal ray to the right).  Only a vertex inside the polygon's bounding box
    // can be inside: below / above it no segment straddles the ray; left of it EVERY straddled segment is crossed - an even
    // number, the polygon is closed - and right of it none (margins as in the per-segment test below).  The sort put the
    // vertices inside the box into waves of their own.
    bool inside = false;
    const bool inb = live && py >= ylo && py < yhi && px >= xlo - xmg && px <= xhi + xmg;
    if (__any(inb)) {
      bool sa = ye[0] > py;
      for (int w = 0; w < NW; ++w) {
        uint32_t sm = 0u;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
          const bool sb = ye[32 * w + k + 1] > py;
          sm |= (sa != sb ? 1u : 0u) << k;
          sa = sb;
        }
        if (!inb) sm = 0u;
        // crossings: xin lies between the segment's end abscissae (the ray straddles it: the parameter is in [0, 1]) up to
        // rounding - |xin_fp - xin| <= 4 eps (|A.x| + |B.x - A.x|) - so a vertex clearly left / right of BOTH ends is
        // decided by comparisons; the division only for the lanes in between (margin 64 eps (|A.x| + |B.x|))
        uint32_t m = sm;
        while (__any(m != 0u)) {
          const bool has = m != 0u;
          const int i = min(32 * w + __ffs((int)m) - 1, np_ - 1);
          m &= m - 1u;
          const double2 A = poly[i], Bp = poly[i + 1 == np_ ? 0 : i + 1];
          const double mg = 1.5e-14 * (fabs(A.x) + fabs(Bp.x)) + 1e-300;
          const bool left = px < fmin(A.x, Bp.x) - mg, right = px > fmax(A.x, Bp.x) + mg;
          if (has && left) inside = !inside;
          const bool amb = has && !left && !right;
          if (__any(amb)) {
            if (amb) {
              const double xin = A.x + (py - A.y) * (Bp.x - A.x) / (Bp.y - A.y);
              if (px < xin) inside = !inside;
            }
          }
        }
      }
    }
    if (live) dist[r] = inside ? 0.0 : sqrt(d2);
  }
  __syncthreads();
  TT_STAMP(12)
  // Stable argsort as a BUCKET sort.  Key of an entry = the leading bits of its (non-negative) distance - 6 bits of
  // clamped exponent (2^-40 .. 2^23; below / above: one bucket each), 5 of mantissa: 2 048 buckets, monotone in the
  // distance.  Histogram by LDS atomics, exclusive scan, scatter by a second round of atomics (any order inside a
  // bucket), then every entry counts the members of ITS bucket that sort before it (distance, then index: unique keys)
  // and is stored at bucket start + count: the order of a stable argsort, exactly.  Buckets hold 1 - 5 entries on the
  // lab meshes: ~4 k cycles against the 38 k of the 1 024-entry bitonic sort (55 LDS stages, 14 barriers) this replaces.
  // A bucket of more than 48 entries (many equal distances, e.g. vertices inside the polygon): the bitonic sort, as before.
  const int off = D.offset[b];
  bool ranked = false;
  {
    constexpr int E0 = 1023 - 40;                          // exponent of bucket group 0 (and everything below it)
    auto bkey = [&](int i_) {
      const uint32_t h = (uint32_t)((unsigned long long)__double_as_longlong(dist[i_]) >> 32);
      const int e = (int)(h >> 20) - E0;
      return e <= 0 ? 0 : (e >= 63 ? 2047 : (e << 5) | (int)((h >> 15) & 31u));
    };
    int* hist = K == 1 ? scanb : reinterpret_cast<int*>(lds_ + 65536 + 20480);   // [2048]
    uint16_t* sidx = reinterpret_cast<uint16_t*>(inv);     // [nrem] entries grouped by bucket (inv is set up after this)
    for (int i = tid; i < 2048; i += TW) hist[i] = 0;
    if (tid == 0) misc[4] = 0;
    __syncthreads();
    for (int i = tid; i < nrem; i += TW) atomicAdd(&hist[bkey(i)], 1);
    __syncthreads();
    for (int i = tid; i < 2048; i += TW)
      if (hist[i] > 48) misc[4] = 1;
    __syncthreads();
    if (misc[4] == 0) {
      ranked = true;
      scan_excl(hist, 2048, part);                         // bucket starts, in place
      __syncthreads();
      for (int i = tid; i < nrem; i += TW) sidx[atomicAdd(&hist[bkey(i)], 1)] = (uint16_t)i;   // hist[k] -> end of bucket k
      __syncthreads();
      for (int i = tid; i < nrem; i += TW) {
        const int k_ = bkey(i);
        const int q0 = k_ == 0 ? 0 : hist[k_ - 1], q1 = hist[k_];
        const double di = dist[i];
        int rank = q0;
        for (int q = q0; q < q1; ++q) {
          const int j_ = sidx[q];
          const double dj = dist[j_];
          rank += (dj < di) || (dj == di && j_ < i);
        }
        order[rank] = (uint16_t)i;
      }
      __syncthreads();
    }
  }
  if (!ranked) {
    int P = 64;
    while (P < nrem) P <<= 1;                       // <= TNV
    for (int i = tid; i < P; i += TW) {
      if (i >= nrem) dist[i] = __builtin_huge_val();
      order[i] = (uint16_t)(i < nrem ? i : 0xFFFF);
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      if (k > 64 && K == 1) __syncthreads();        // (the wave-local stages of the previous k are done everywhere)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P; i += TW) {
          const int x = i ^ j;
          if (x > i) {
            const bool up = (i & k) == 0;
            const double a_ = dist[i], c_ = dist[x];
            const uint16_t ia = order[i], ic = order[x];
            const bool after = a_ > c_ || (a_ == c_ && ia > ic);      // the pair (a, ia) sorts behind (c, ic)
            if (after == up) {
              dist[i] = c_;
              dist[x] = a_;
              order[i] = ic;
              order[x] = ia;
            }
          }
        }
        if (j >= 64 || K != 1) __syncthreads();   // (K != 1: the tables are in global memory - no in-order guarantee without a fence)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  }
  TT_STAMP(13)
  for (int v = tid; v < TNV; v += TW) inv[v] = -1;
  __syncthreads();
  int nsel = nrem - off;
  nsel = nsel > D.N ? D.N : (nsel < 0 ? 0 : nsel);
  {
    int32_t* nc = D.n_closest + Bq * D.N;
    int32_t* cm = D.coord_map + Bq * D.N;
    for (int i = tid; i < D.N; i += TW) {
      if (i < nsel) {
        const int r = order[off + i];
        nc[i] = r;
        cm[i] = remv[r];
        inv[remv[r]] = (int16_t)i;
      } else {
        nc[i] = 0;
        cm[i] = 0;
      }
    }
  }
  if (tid == 0) D.nsel[b] = nsel;
  __syncthreads();
  TT_STAMP(4)
  // ================= state graph: cells whose three vertices are all selected, in cell order
  for (int t = tid; t < TNS; t += TW) {
    int f = 0;
    if (t < nt) f = (inv[tri[3 * t]] >= 0 && inv[tri[3 * t + 1]] >= 0 && inv[tri[3 * t + 2]] >= 0) ? 1 : 0;
    scanb[t] = f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
  __syncthreads();
  const int ngood = scan_excl(scanb, TNS, part);
  if (3 * ngood > D.EMAX) {
    if (tid == 0) status[b] = -3;
    return;
  }
  {
    int32_t* es = D.edge_src + Bq * D.EMAX;
    int32_t* ed = D.edge_dst + Bq * D.EMAX;
    double* el = D.edge_len + Bq * D.EMAX;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid * PER + i;
      if (t < nt && flg.get(i)) {
        const int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
        const int vs[3] = {v0, v1, v2}, id[3] = {inv[v0], inv[v1], inv[v2]};
        const int pa[3] = {0, 0, 1}, pb[3] = {1, 2, 2};
        const int base = 3 * scanb[t];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          es[base + q] = id[pa[q]];
          ed[base + q] = id[pb[q]];
          const double dx = X[vs[pa[q]]].x - X[vs[pb[q]]].x, dy = X[vs[pa[q]]].y - X[vs[pb[q]]].y;
          el[base + q] = sqrt(dx * dx + dy * dy);
        }
      }
    }
  }
  if (tid == 0) D.nedges[b] = 3 * ngood;
  TT_STAMP(5)
  }
  if (!has_ipcs) return;
  __syncthreads();

  // ================= index data of the matrix-free IPCS path (ipcs_topology_one of the host engine)
  int32_t* scat = O.mf_scat + Bq * 6 * D.NT;
  int8_t* cof = O.cell_outflow + Bq * D.NT;
  uint8_t* fl = O.bcu_flag + Bq * D.NP;
  double* gx = O.bcu_gx + Bq * D.NP;
  uint8_t* pf = O.bcp_flag + Bq * D.NV;
  for (int t = tid; t < D.NT; t += TW) cof[t] = -1;
  for (int i = tid; i < D.NP; i += TW) {
    fl[i] = 0;
    gx[i] = 0.0;
  }
  for (int i = tid; i < D.NV; i += TW) pf[i] = 0;
  __syncthreads();
  const double H = top - bot, Um = 1.5;
  // Dirichlet data, list order [inlet, airfoil, walls]: later wins on shared dofs -> three ordered passes
  for (int w = 0; w < 3; ++w) {
    const int want = w == 0 ? 2 : (w == 1 ? 1 : 0);
    for (int e = tid; e < ne; e += TW) {
      if ((eflag[e] >> 4) != want + 1) continue;
      const int dofs[3] = {ea[e], eb[e], nv + e};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int dq = dofs[q];
        fl[dq] = 1;
        if (want == 2) {
          const double y = q < 2 ? X[dq].y : 0.5 * (X[ea[e]].y + X[eb[e]].y);
          gx[dq] = -4.0 * Um * (y - bot) * (y - top) / H / H;
        } else {
          gx[dq] = 0.0;
        }
      }
    }
    __syncthreads();
  }
  TT_STAMP(7)
  // outflow facets: pressure Dirichlet vertices, cell -> local facet, entries (row, col, src) of the facet term
  for (int e = tid; e < TNS; e += TW) scanb[e] = (e < ne && (eflag[e] >> 4) == 4) ? 1 : 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
  __syncthreads();
  const int nof = scan_excl(scanb, TNS, part);
  const int nent = 18 * nof;
  if (nent > O.NBE || nent > 2 * TNV) {
    if (tid == 0) status[b] = -5;
    return;
  }
  // entry keys (aliasing dist / poly: both dead): key = row << RSH | col << CSH, src kept beside it (12-bit dof ids and 32-bit
  // keys for K = 1, 14-bit ids and 64-bit keys for the large-mesh instance)
  using EKey = typename std::conditional<K == 1, uint32_t, uint64_t>::type;
  constexpr int RSH = K == 1 ? 20 : 40, CSH = K == 1 ? 8 : 20;
  constexpr EKey CMASK = K == 1 ? (EKey)0xFFF : (EKey)0xFFFFF;
  EKey* ekey = reinterpret_cast<EKey*>(dist);              // [nent] (<= 1024 entries: 8 KB of the 8 TNV bytes of `dist`)
  int32_t* esrc = reinterpret_cast<int32_t*>(poly);        // [nent] -> spills into cntd for nent > 1024: keep it small
  if (nent > 1024) {
    if (tid == 0) status[b] = -5;
    return;
  }
  // (one thread per ENTRY: as one thread per facet the 36 reads of the cell's dofs were a chain of L2 round trips)
  constexpr int ROFE = RSEG + (2 * TNP + 16) * 4 + 8192;     // (behind cntd and `fill`; K = 1: 61 472)
  int* ofe = reinterpret_cast<int*>(R + ROFE);               // [nof] outflow edges in edge order
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid * PER + i;
    if (e < ne && flg.get(i)) {
      pf[ea[e]] = 1;
      pf[eb[e]] = 1;
      cof[eown[e] / 3] = (int8_t)(eown[e] % 3);
      ofe[scanb[e]] = e;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < nent; idx += TW) {
    const int f = idx / 18, rem = idx - 18 * f, q = rem / 6, j = rem - 6 * q;
    const int e = ofe[f];
    const int c = eown[e] / 3, k = eown[e] % 3;
    const int rowl = q == 0 ? (k == 0 ? 1 : 0) : (q == 1 ? (k == 2 ? 1 : 2) : 3 + k);
    const int row = cd[rowl * D.NT + c], col = cd[j * D.NT + c];
    ekey[idx] = ((EKey)row << RSH) | ((EKey)col << CSH);  // (src breaks ties below)
    esrc[idx] = c * 36 + rowl * 6 + j;
  }
  __syncthreads();
  TT_STAMP(14)
  {
    int32_t* bo_rows = O.bo_rows + Bq * O.NBO;
    int32_t* bo_ptr = O.bo_ptr + Bq * (O.NBO + 1);
    int32_t* bo_col = O.bo_col + Bq * O.NBE;
    int32_t* bo_src = O.bo_src + Bq * O.NBE;
    int* lrow = reinterpret_cast<int*>(R + ROFE + 1024);       // [nbo] rows of the outflow list (behind `ofe`)
    // rank by counting over (row, col, src); then rows = runs of equal row
    // (four lanes per entry, a quarter of the list each: one lane per entry left most of the workgroup idle on a loop of
    //  ~450 steps)
    {
      const int QE = (nent + 3) >> 2;
      for (int it0 = 0; it0 < 4 * nent; it0 += TW) {
        const int it = it0 + tid, t = min(it >> 2, nent - 1), part_ = it & 3;
        const EKey kt = ekey[t];
        const int st = esrc[t];
        const int q0 = part_ * QE, q1 = min(nent, q0 + QE);
        int rank = 0;
#pragma unroll 8
        for (int q = q0; q < q1; ++q) rank += (ekey[q] < kt) || (ekey[q] == kt && esrc[q] < st);
        rank += __shfl_xor(rank, 1, 64);
        rank += __shfl_xor(rank, 2, 64);
        if (part_ == 0 && (it >> 2) < nent) {
          bo_col[rank] = (int32_t)((kt >> CSH) & CMASK);
          bo_src[rank] = st;
          cntd[rank] = (int32_t)(kt >> RSH);  // row of the sorted entry
        }
      }
    }
    __syncthreads();
    TT_STAMP(15)
    for (int t = tid; t < TNS; t += TW) scanb[t] = (t < nent && (t == 0 || cntd[t] != cntd[t - 1])) ? 1 : 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) flg.set(i, scanb[tid * PER + i] != 0);
    __syncthreads();
    const int nbo = scan_excl(scanb, TNS, part);
    if (nbo > O.NBO) {
      if (tid == 0) status[b] = -5;
      return;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid * PER + i;
      if (t < nent && flg.get(i)) {
        bo_rows[scanb[t]] = cntd[t];
        lrow[scanb[t]] = cntd[t];
        bo_ptr[scanb[t]] = t;
      }
    }
    if (tid == 0) {
      bo_ptr[nbo] = nent;
      O.nbo[b] = nbo;
      misc[0] = 0;
    }
    __syncthreads();
    // at most two outflow rows per row-owner thread of the 512-thread kernels (mode 3: meshes of the K = 1 instance only)
    for (int t = tid; t < (K == 1 ? nbo : 0); t += TW) {     // (from the LDS copy of the row list: as a loop of global loads of the list
      int same = 0;                           //  just written it was ~nbo dependent L2 round trips, most of this phase)
      const int mine = lrow[t] % 512;
#pragma unroll 8
      for (int q = 0; q < nbo; ++q) same += (lrow[q] % 512) == mine;
      if (same > 2) misc[0] = 1;
    }
    __syncthreads();
    if (misc[0]) {
      if (tid == 0) status[b] = -4;
      return;
    }
  }
  TT_STAMP(8)
  // packed per-triangle metadata (mode 3; the large-mesh instance feeds mode 5, which reads cell_dofs / cell_outflow themselves)
  if (K == 1)
    for (int i = 0; i < 6; ++i)
      for (int t = tid; t < nt; t += TW) scat[i * D.NT + t] = cd[i * D.NT + t] | (i == 0 ? ((int32_t)(cof[t] + 1) << 28) : 0);
  // dof <- element-slot gathers, ascending slots: count, scan, unordered fill and per-dof sort of the (short) lists in an
  // LDS staging array (slot ids fit 16 bits), then one coalesced copy out.  (Filling and insertion-sorting the lists
  // in global memory was 78 k cycles of dependent L2 round trips for the two lists.)
  using stage_t = typename CP::stage_t;
  stage_t* stage = K > 4 ? reinterpret_cast<stage_t*>(smem + CP::BASE) : reinterpret_cast<stage_t*>(R);   // [nl * nt] <= 6 TNT entries = 24 KB (blist .. cntd are dead)
  static_assert(K > 4 || 6 * TNT * 2 <= RSEG + (TNP + 8) * 4, "gather staging fits in front of `fill`");
  auto gather = [&](int nl, int ndof, int32_t* gptr, int32_t* gsrc) {
    for (int i = tid; i < TNS; i += TW) scanb[i] = 0;
    __syncthreads();
    for (int i = 0; i < nl; ++i)
      for (int t = tid; t < nt; t += TW) atomicAdd(&scanb[cd[i * D.NT + t]], 1);
    __syncthreads();
    const int tot = scan_excl(scanb, TNS, part);
    (void)tot;
    for (int i = tid; i <= ndof; i += TW) {
      gptr[i] = i < ndof ? scanb[i] : nl * nt;
      if (i < ndof) fill[i] = 0;
    }
    __syncthreads();
    for (int i = 0; i < nl; ++i)
      for (int t = tid; t < nt; t += TW) {
        const int dof = cd[i * D.NT + t];
        stage[scanb[dof] + atomicAdd(&fill[dof], 1)] = (stage_t)(t * nl + i);   // slot = t * nl + i
      }
    __syncthreads();
    for (int i = tid; i < ndof; i += TW) {
      const int q0 = scanb[i], q1 = q0 + fill[i];
      for (int a_ = q0 + 1; a_ < q1; ++a_) {
        const stage_t w = stage[a_];
        int j = a_ - 1;
        while (j >= q0 && stage[j] > w) {
          stage[j + 1] = stage[j];
          --j;
        }
        stage[j + 1] = w;
      }
    }
    __syncthreads();
    for (int q = tid; q < nl * nt; q += TW) gsrc[q] = stage[q];
    if (nl == 6) {
      // The P1 lists (vertex <- (cell, local vertex) slots, slot = 3 t + i) ARE the vertex rows of the P2 lists (slot =
      // 6 t + i, i < 3): the vertex dofs come first, a vertex has one slot per incident cell, and both lists ascend with
      // the cell.  They are written from the sorted staging array here instead of by a count / scan / fill / sort pass of
      // their own (16 k of the flow variant's 128 k cycles); the same arrays, bit for bit.
      int32_t* g1p_ = O.g1_ptr + Bq * (D.NV + 1);
      int32_t* g1s_ = O.g1_src + Bq * 3 * D.NT;
      for (int i = tid; i <= nv; i += TW) g1p_[i] = i < nv ? scanb[i] : 3 * nt;
      for (int q = tid; q < 3 * nt; q += TW) {
        const int w = stage[q];
        g1s_[q] = (w / 6) * 3 + (w - (w / 6) * 6);
      }
    }
    __syncthreads();
  };
  TT_STAMP(9)
  TT_STAMP(10)
  gather(6, n2, O.g2_ptr + Bq * (D.NP + 1), O.g2_src + Bq * 6 * D.NT);
  TT_STAMP(11)
  // SELL-64 pattern of the P1 Laplacian: row = {vertex} + neighbours, ascending; slice width = longest row.
  // The neighbour lists come from the edge arrays, all in LDS: degree by atomics, scan, unordered fill, then every row
  // ranks its (distinct) entries by counting.  (The first version collected a row's columns from its incident cells
  // through the g1 lists and the cell array in GLOBAL memory into a dynamically indexed local array: chains of dependent
  // L2 round trips and scratch traffic, 95 k of the flow variant's 570 k cycles.)
  uint16_t* adjl = reinterpret_cast<uint16_t*>(R);           // [2 ne] <= 2 TNE entries (blist .. poly are dead)
  int* rptr = cntd;                                          // [nv + 1] row pointers of adjl (cntd is dead)
  int* rcur = cntd + TNV + 8;                                // [nv] fill cursors
  for (int i = tid; i < TNS; i += TW) scanb[i] = 0;
  __syncthreads();
  for (int e = tid; e < ne; e += TW) {
    atomicAdd(&scanb[ea[e]], 1);
    atomicAdd(&scanb[eb[e]], 1);
  }
  __syncthreads();
  for (int i = tid; i < nv; i += TW) {
    fill[i] = scanb[i] + 1;   // row lengths (the diagonal + the incident edges)
    rcur[i] = 0;
  }
  __syncthreads();
  scan_excl(scanb, TNS, part);
  for (int i = tid; i <= nv; i += TW) rptr[i] = i < nv ? scanb[i] : 2 * ne;
  __syncthreads();
  for (int e = tid; e < ne; e += TW) {
    const int a_ = ea[e], b_ = eb[e];
    adjl[rptr[a_] + atomicAdd(&rcur[a_], 1)] = (uint16_t)b_;
    adjl[rptr[b_] + atomicAdd(&rcur[b_], 1)] = (uint16_t)a_;
  }
  __syncthreads();
  {
    int32_t* so = O.sl1_off + Bq * (D.NV / 64 + 2);
    int32_t* sc = O.sl1_col + Bq * O.NSE1;
    const int ns = (nv + 63) / 64;
    // slice widths -> offsets (one wave per slice: the longest of its 64 rows by a shuffle reduction; as one THREAD per
    // slice walking 64 rows the 14 slices of a mesh cost 64 dependent LDS reads each)
    for (int s_ = tid; s_ < TNS; s_ += TW) scanb[s_] = 0;
    __syncthreads();
    for (int s_ = tid >> 6; s_ < ns; s_ += TW / 64) {
      const int r = 64 * s_ + (tid & 63);
      int w = r < nv ? fill[r] : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o, 64));
      if ((tid & 63) == 0) scanb[s_] = 64 * w;
    }
    __syncthreads();
    const int total = scan_excl(scanb, TNS, part);
    if (total > O.NSE1) {
      if (tid == 0) status[b] = -6;
      return;
    }
    for (int s_ = tid; s_ <= ns; s_ += TW) so[s_] = s_ < ns ? scanb[s_] : total;
    for (int r = tid; r < 64 * ns; r += TW) {
      const int s_ = r >> 6, l = r & 63;
      const int base = scanb[s_], w = ((s_ + 1 < ns ? scanb[s_ + 1] : total) - base) >> 6;
      const int rr = min(r, nv - 1);
      int len = 0;
      if (r < nv) {
        const int q0 = rptr[r], q1 = rptr[r + 1];
        len = q1 - q0 + 1;
        int below = 0;                                       // neighbours below the diagonal
        for (int q = q0; q < q1; ++q) {
          const int v = adjl[q];
          int rank = v > r ? 1 : 0;                          // (the diagonal entry)
          for (int q2 = q0; q2 < q1; ++q2) rank += adjl[q2] < v ? 1 : 0;
          below += v < r ? 1 : 0;
          sc[base + 64 * rank + l] = v;
        }
        sc[base + 64 * below + l] = r;
      }
      for (int j = len; j < w; ++j) sc[base + 64 * j + l] = rr;
    }
  }
  TT_STAMP(6)
}

}  // namespace mdq_topo

extern "C" int64_t mdq_env_topology_workspace_bytes(const mdq_env_topo_desc* d) {
  if (!d || d->B <= 0) return 0;
  if (d->NV <= mdq_topo::TNV && d->NT <= mdq_topo::TNT && d->NP <= mdq_topo::TNP) return 0;     // every table in LDS
  using C4 = mdq_topo::TCap<4>;
  using C16 = mdq_topo::TCap<16>;
  if (d->NV > C16::NV || d->NT > C16::NT || d->NP > C16::NP) return -1;                          // beyond the kernels
  if (d->NV > C4::NV || d->NT > C4::NT || d->NP > C4::NP) return (int64_t)((C16::BYTES + 255) & ~(size_t)255) * d->B;
  return (int64_t)((C4::BYTES + 255) & ~(size_t)255) * d->B;
}

extern "C" int mdq_env_topology(const mdq_env_topo_desc* d, void* stream, int32_t* status) {
  if (!d || d->B <= 0 || !status) return mdq_set_error("mdq_env_topology: bad arguments");
  {
    using C4_ = mdq_topo::TCap<4>;
    const bool huge_ = d->NV > C4_::NV || d->NT > C4_::NT || d->NP > C4_::NP;
    if (d->npoly > (huge_ ? 2 : 1) * mdq_topo::TNPOLY)
      return mdq_set_error("mdq_env_topology: more than 256 polygon points (512 for the 16384-vertex instance)");
  }
  mdq_ipcs_topo_out o = {};
  if (d->ipcs) o = *d->ipcs;
  mdq_topo_handover h = {};
  if (d->handover) h = *d->handover;
  if (d->NV <= mdq_topo::TNV && d->NT <= mdq_topo::TNT && d->NP <= mdq_topo::TNP) {
    const size_t lds = mdq_topo::TCap<1>::BYTES;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_topo::topology_kernel<1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return mdq_set_error("hipFuncSetAttribute(topology_kernel) failed");
    hipLaunchKernelGGL(mdq_topo::topology_kernel<1>, dim3(d->B), dim3(mdq_topo::TW), lds, (hipStream_t)stream, *d, o,
                       d->ipcs ? 1 : 0, status, nullptr, h);
  } else {
    using C4 = mdq_topo::TCap<4>;
    using C16 = mdq_topo::TCap<16>;
    if (d->NV > C16::NV || d->NT > C16::NT || d->NP > C16::NP)
      return mdq_set_error("mdq_env_topology: capacity above 16384 vertices / 32768 triangles / 65536 P2 dofs");
    const bool huge = d->NV > C4::NV || d->NT > C4::NT || d->NP > C4::NP;
    // the large-mesh instance: tables in the CALLER's workspace (global memory, one slab per mesh)
    if (!d->workspace || d->workspace_bytes < mdq_env_topology_workspace_bytes(d) || (reinterpret_cast<uintptr_t>(d->workspace) & 15))
      return mdq_set_error("mdq_env_topology: workspace missing, too small or not 16-byte aligned (mdq_env_topology_workspace_bytes)");
    unsigned char* slab = static_cast<unsigned char*>(d->workspace);
    const size_t lds4 = 2 * sizeof(uint32_t) * 16384;      // the edge hash, then the coordinates (see the kernel)
    static const hipError_t attr4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_topo::topology_kernel<4>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    if (attr4 != hipSuccess) return mdq_set_error("hipFuncSetAttribute(topology_kernel<4>) failed");
    if (huge) {
      // the 16 384-vertex instance (round 6): every table incl. the edge hash and the coordinates on the slab; the LDS keeps the
      // small tables of the K = 4 layout (upper 64 KB .. 128 KB: polygon tables, hash sets of the boundary values, histograms)
      static const hipError_t attr16 = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_topo::topology_kernel<16>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      if (attr16 != hipSuccess) return mdq_set_error("hipFuncSetAttribute(topology_kernel<16>) failed");
      hipLaunchKernelGGL(mdq_topo::topology_kernel<16>, dim3(d->B), dim3(mdq_topo::TW), lds4, (hipStream_t)stream, *d, o,
                         d->ipcs ? 1 : 0, status, slab, h);
    } else
    hipLaunchKernelGGL(mdq_topo::topology_kernel<4>, dim3(d->B), dim3(mdq_topo::TW), lds4, (hipStream_t)stream, *d, o,
                       d->ipcs ? 1 : 0, status, slab, h);
  }
  if (hipGetLastError() != hipSuccess) return mdq_set_error("topology_kernel launch failed");
  return 0;
}
