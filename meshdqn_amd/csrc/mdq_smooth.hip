// DOLFIN `Mesh.smooth(n)` (flow_solver.py:65-67, 236-237) on the GPU for B meshes at once.
//
// The algorithm is a Gauss-Seidel sweep over the interior vertices IN INDEX ORDER, repeated n times:
//   c = mean of the edge-connected neighbours, r_min = min distance from the vertex to the lines through the
//   opposite edges of its cells, p <- p + min(|c-p|, r_min/2) (c-p)/|c-p|.
// Inside a sweep vertex v needs the new value of its neighbours w < v and the old value of w > v, so the updates of
// one sweep form a dependency graph whose depth is the longest ascending path of the mesh numbering: 113 levels
// for the 694 interior vertices of ys930 (the ring of consecutively numbered vertices around the airfoil), 6 updates
// per level on average.  That is a LATENCY problem (34 700 updates, ~5 500 of them in sequence), and one wavefront
// runs it faster than eight did: a first version polled per-vertex sweep counters from 64 queue-owning lane groups
// in 8 waves (dataflow); its waves were busy a quarter of the time, every hand-off paid a poll, and an update waited
// on average several thousand cycles for its group to reach it (in-order queues).  Now:
//   * the setup phase (256 threads) LIST-schedules one sweep: a vertex is ready when its lower-numbered interior
//     neighbours are scheduled, every pass takes the 8 ready vertices with the smallest indices (119 passes per sweep
//     on ys930; level by level it was 141, the critical path is 113);
//   * ONE wave walks the passes, sweep after sweep, 8 lanes per update (one incident cell each): no flags, no
//     polling, no barriers - the LDS serves a wave's operations in order, so the position loads of a pass see the
//     stores of the pass before;
//   * a vertex record holds (x, y, y, x): odd lanes read the swapped pair, so that the first stage of the centroid
//     reduction exchanges the component a lane does NOT keep - x and y are summed over the 8 lanes with three
//     exchanges instead of six (even lanes end with the x sum, odd lanes with the y sum);
//   * the step limit |c - p| <= r_min / 2 is DECIDED in fp32 with a safety margin (squared altitudes by one
//     v_rcp_f32, 8-lane minimum by three integer minima on the bit patterns); an undecided or limited update runs the
//     exact fp64 path (exact_update, any degree);
//   * the first three sweeps are CAREFUL (decision in front of the store): right after a vertex removal the cavity's
//     neighbours take limited steps in most launches.  The rest are SPECULATIVE: the position is stored first and the
//     test of a pass runs behind the position loads of the next pass; the records are checkpointed in the coordinate
//     array, and a sweep with an undecided update is redone careful from the checkpoint (doubling back-off);
//   * vertices of more than 8 cells (a few after removals) sit in listed pass pairs that run careful between
//     speculative segments (a branch for them inside the speculative loop body cost 20 %);
//   * the metadata of a pass (record addresses of each lane's cell) is prefetched up to three passes ahead.
// Updates of one pass are independent and every reduction is group-local in a fixed lane order: results are
// bitwise reproducible and agree with the sequential host loop to round-off (different association).
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_smoothing {
// diagnostics (mdq_smooth_stats): [s] = speculative sweeps s that were abandoned for a careful redo
__device__ unsigned long long g_abandoned[64];

constexpr int SNV = 1024;      // vertex capacity (record offsets must fit 16 bits)
constexpr int SNT = 2048;      // triangle capacity (cell id must fit 12 bits)
constexpr int SWG = 256;       // threads per workgroup (setup phase; the sweeps run in wave 0)
constexpr int GRP = 8;         // lanes per vertex update
constexpr int PER = SNV / SWG; // entries per thread in the setup scans
constexpr int REC = 32;        // bytes per vertex record: x, y, y, x
constexpr int ZREC = SNV * REC;        // all-zero record: the unused lanes of a vertex with fewer than 8 cells, empty pass slots
constexpr int AREC = (SNV + 1) * REC;  // (0, -1000) and
constexpr int CREC = (SNV + 2) * REC;  // (1, -1000): the far-away edge that keeps an empty pass slot on the fast path
constexpr int ROW = 48;        // bytes per metadata row of an interior vertex: 8 lane words a | c << 16 (record addresses of the
                               // lane's cell), 1 / (2 degree), record address of the vertex, lower limit of |c - p|^2 for the fast path

// LDS carve-up (one array: the records start at LDS address 0, so record offsets ARE ds_* addresses)
constexpr int OFF_REC = 0;
constexpr int OFF_ROW = OFF_REC + (SNV + 3) * REC;
constexpr int OFF_PT = OFF_ROW + (SNV + 1) * ROW;           // passes: 8 row offsets (uint16) each; even count + 2 wrapped around
constexpr int OFF_PTR = OFF_PT + (SNV + 4) * GRP * 2;
constexpr int OFF_CNT = OFF_PTR + (SNV + 1) * 4 + 12;
constexpr int OFF_INC = OFF_CNT + SNV * 4;
constexpr int OFF_IVERT = OFF_INC + 3 * SNT * 4;
constexpr int OFF_RK = OFF_IVERT + SNV * 2;                 // interior rank of a vertex (0xFFFF: not interior)
constexpr int OFF_DEG = OFF_RK + (SNV + 8) * 2;             // unscheduled lower neighbours per interior rank (x 2)
constexpr int OFF_R2K = OFF_DEG + SNV * 4;                  // 1 / (2 k)
constexpr int OFF_RB = OFF_R2K + 32 * 8;                    // ready bitmap over the interior ranks
constexpr int OFF_PART = OFF_RB + 2 * (SNV / 64) * 8;
constexpr int LDS_BYTES = OFF_PART + SWG * 4;
constexpr int MAXBP = SWG - 2;                               // pass pairs with a vertex of more than 8 cells (listed behind part[1])
static_assert(OFF_ROW % 16 == 0 && OFF_PT % 16 == 0 && OFF_PTR % 16 == 0 && OFF_CNT % 16 == 0 && OFF_R2K % 8 == 0 &&
                  OFF_RB % 8 == 0 && OFF_DEG % 4 == 0, "LDS alignment");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(SNV * GRP * 4 <= (SNV + 3) * REC, "scheduler scratch inside the record area");

typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) d2 lds_d2;
typedef __attribute__((address_space(3))) u4 lds_u4;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) uint16_t lds_u16;

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
  return v;
}
__device__ __forceinline__ double grp_min(double v) {
  v = fmin(v, dpp8<0xB1>(v));
  v = fmin(v, dpp8<0x4E>(v));
  v = fmin(v, dpp8<0x141>(v));
  return v;
}
// 1/sqrt(x) and 1/x to double precision (~1 ulp): hardware estimate + two Newton steps (exact path only)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
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

// exact fp64 update of vertex `v` by the 8 lanes of its group (any degree, limited steps): returns the new value of
// the lane's component (lane 0: x, lane 1: y) in `pn` and whether the vertex moves at all
__device__ __forceinline__ bool exact_update(const lds_u8* recb, const lds_i32* ptr, const lds_u32* inc, int v, int l,
                                             double& pn) {
#pragma clang fp contract(off)
  const double EPS = 3.0e-16;
  const int q0 = ptr[v], k = ptr[v + 1] - q0;
  const d2 p = *reinterpret_cast<const lds_d2*>(recb + v * REC);
  double sx = 0.0, sy = 0.0, rm = 1e300;
  for (int q = l; q < k; q += GRP) {
    const uint32_t w = inc[q0 + q];
    const int a = w & 0x3FF, c = (w >> 10) & 0x3FF;
    const d2 pa = *reinterpret_cast<const lds_d2*>(recb + a * REC);
    const d2 pc = *reinterpret_cast<const lds_d2*>(recb + c * REC);
    sx += pa.x + pc.x;
    sy += pa.y + pc.y;
    const double tx = pc.x - pa.x, ty = pc.y - pa.y;
    const double cr = ty * (p.x - pa.x) - tx * (p.y - pa.y);
    rm = fmin(rm, cr * cr * rcp_nr(tx * tx + ty * ty));   // SQUARED distance to the line through the opposite edge
  }
  sx = grp_sum(sx);
  sy = grp_sum(sy);
  rm = grp_min(rm);
  const double r2k = 1.0 / (2.0 * k);
  const double pc_ = l == 0 ? p.x : p.y;
  const double dc_ = (l == 0 ? sx : sy) * r2k - pc_;    // lane 0: dx, lane 1: dy
  const double dother = dpp8<0xB1>(dc_);
  const double q2 = dc_ * dc_ + dother * dother;
  pn = pc_;
  if (!(q2 >= EPS * EPS && q2 > 0.0)) return false;     // |c - p| < DOLFIN_EPS: the vertex stays
  pn = pc_ + dc_;                                       // |c - p| <= r_min / 2: to the centroid itself
  if (0.25 * rm < q2) {                                 // limited step: needs the lengths
    const double ir = rsqrt_nr(q2);
    const double rmin = rm * rsqrt_nr(rm);
    pn = pc_ + (0.5 * rmin * ir) * dc_;
  }
  return true;
}

// one environment (b) by the whole workgroup
__device__ __forceinline__ void smooth_env(unsigned char* lds, const int b, int NV, int NT, double* coords, const int32_t* cells,
                                           const int32_t* nv_, const int32_t* nt_, const int32_t* iters_, int cap,
                                           long long* trace) {
  unsigned char* recb = lds + OFF_REC;
  unsigned char* rows = lds + OFF_ROW;
  uint16_t* passtab = reinterpret_cast<uint16_t*>(lds + OFF_PT);
  int* ptr = reinterpret_cast<int*>(lds + OFF_PTR);
  int* cnt = reinterpret_cast<int*>(lds + OFF_CNT);
  uint32_t* inc = reinterpret_cast<uint32_t*>(lds + OFF_INC);   // a | b << 10 | cell << 20 of every (vertex, incident cell), grouped by vertex, ascending cell
  uint16_t* ivert = reinterpret_cast<uint16_t*>(lds + OFF_IVERT);   // interior vertices in index order
  uint16_t* rk = reinterpret_cast<uint16_t*>(lds + OFF_RK);
  int* indeg2 = reinterpret_cast<int*>(lds + OFF_DEG);
  unsigned long long* rb = reinterpret_cast<unsigned long long*>(lds + OFF_RB);
  unsigned long long* bigdeg = rb + SNV / 64;                   // interior ranks of degree > 8
  unsigned char* hwt = recb;   // scheduler scratch in the record area (the records are loaded behind the scheduler):
                               // per interior rank and cell the interior ranks of its higher-numbered vertices
  int* part = reinterpret_cast<int*>(lds + OFF_PART);
  const int tid = threadIdx.x;
  const int iters = cap > 0 ? min(iters_[b], cap) : iters_[b];   // (cap: only the first sweeps)
  if (iters <= 0) return;
#ifdef MDQ_SMOOTH_TRACE
  int phase_ = 0;   // setup phase stamps of environment 0 in the slots of sweep 63
#define MDQ_SMOOTH_PHASE() if (trace && b == 0 && tid == 0) trace[2 * (63 * SNV + phase_++)] = clock64();
#else
#define MDQ_SMOOTH_PHASE()
#endif
  MDQ_SMOOTH_PHASE()
  const int nv = nv_[b], nt = nt_[b];
  double2* x = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  const int32_t* tri = cells + (int64_t)b * NT * 3;
  for (int v = tid; v < SNV; v += SWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += SWG)
    for (int k = 0; k < 3; ++k) atomicAdd(&cnt[tri[3 * t + k]], 1);
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 1: counted */
  scan_inclusive(cnt, part);
  for (int v = tid; v < SNV; v += SWG) ptr[v + 1] = cnt[v];
  if (tid == 0) ptr[0] = 0;
  __syncthreads();
  // cell lists in arrival order (scratch: the metadata rows are not built yet), owner of every entry
  uint32_t* tmp = reinterpret_cast<uint32_t*>(rows);
  uint16_t* own = reinterpret_cast<uint16_t*>(rows + 3 * SNT * 4);
  static_assert(3 * SNT * 6 <= (SNV + 1) * ROW, "scratch inside the row area");
  for (int v = tid; v < SNV; v += SWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += SWG) {
    const int vs[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k], a = vs[(k + 1) % 3], c = vs[(k + 2) % 3];
      const int q = ptr[v] + atomicAdd(&cnt[v], 1);
      tmp[q] = (uint32_t)a | ((uint32_t)c << 10) | ((uint32_t)t << 20);
      own[q] = (uint16_t)v;
    }
  }
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 2: cell lists filled */
  for (int v = tid; v < SNV; v += SWG) cnt[v] = 0;   // now: 1 = a neighbour of the vertex is not seen exactly twice
  __syncthreads();
  // one thread per (vertex, incident cell): rank of the cell among the vertex's cells (ascending cell id: the fixed
  // order the lanes of an update take them in), boundary test of its two other vertices
  for (int e = tid; e < 3 * nt; e += SWG) {
    const int v = own[e], q0 = ptr[v], q1 = ptr[v + 1];
    const uint32_t my = tmp[e], a = my & 0x3FF, c = (my >> 10) & 0x3FF;
    // entries beyond the vertex's own: a sentinel (the vertex itself as both other vertices - never a neighbour of
    // itself - and the largest cell id), so that the counting below needs no range test
    const uint32_t sent = (uint32_t)v | ((uint32_t)v << 10) | 0xFFF00000u;
    uint32_t o[GRP];
#pragma unroll
    for (int j = 0; j < GRP; ++j) o[j] = tmp[min(q0 + j, q1 - 1)];   // (all loads in flight together)
    uint32_t rank = 0, sa = 0, sc = 0;   // differences are counted: xor + min(., 1) + add, no compare masks
    const uint32_t mykey = my >> 20;
#pragma unroll
    for (int j = 0; j < GRP; ++j) {
      const uint32_t oj = q0 + j < q1 ? o[j] : sent, oa = oj & 0x3FF, oc = (oj >> 10) & 0x3FF;
      rank += min(mykey - min(oj >> 20, mykey), 1u);   // 1 if the other cell id is smaller
      sa += min(oa ^ a, 1u) + min(oc ^ a, 1u);
      sc += min(oa ^ c, 1u) + min(oc ^ c, 1u);
    }
    sa = 2 * GRP - sa;   // matches = slots - differences
    sc = 2 * GRP - sc;
    for (int j = q0 + GRP; j < q1; ++j) {   // degree > 8
      const uint32_t oj = tmp[j], oa = oj & 0x3FF, oc = (oj >> 10) & 0x3FF;
      rank += (oj >> 20) < mykey;
      sa += (oa == a) + (oc == a);
      sc += (oa == c) + (oc == c);
    }
    inc[q0 + rank] = my;
    if (sa != 2 || sc != 2) cnt[v] = 1;
  }
  __syncthreads();
  for (int v = tid; v < SNV; v += SWG) {
    const bool interior = v < nv && ptr[v + 1] > ptr[v] && cnt[v] == 0;
    cnt[v] = interior ? 1 : 0;
  }
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 3: sorted, interior test, records */
  scan_inclusive(cnt, part);
  const int n_int = cnt[SNV - 1];
  __syncthreads();
  for (int v = tid; v < SNV + 8; v += SWG) {
    const bool interior = v < nv && cnt[v] != (v ? cnt[v - 1] : 0);
    rk[v] = interior ? (uint16_t)(cnt[v] - 1) : (uint16_t)0xFFFF;
    if (interior) ivert[cnt[v] - 1] = (uint16_t)v;
  }
  double* r2ktab = reinterpret_cast<double*>(lds + OFF_R2K);   // 1 / (2 k), k < 32
  if (tid < 32) r2ktab[tid] = tid ? 1.0 / (2.0 * tid) : 0.0;
  if (tid < SNV / 64) rb[tid] = bigdeg[tid] = 0ull;
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 4: interior ranks */
  // metadata rows.  Row n_int is the empty pass slot: vertex and cells on the zero record except lane 0, whose cell
  // is the far-away edge (a finite altitude keeps the slot on the fast path; it stores zeros into the zero record)
  for (int r = tid; r <= n_int; r += SWG) {
    uint32_t w[GRP], wv = (uint32_t)ZREC;
    double r2k = 0.0;
    float thr = -1.0f;
#pragma unroll
    for (int l = 0; l < GRP; ++l) w[l] = ((uint32_t)ZREC + (l & 1) * 16) * 0x10001u;
    if (r < n_int) {
      const int v = ivert[r], q0 = ptr[v], k = ptr[v + 1] - q0;
#pragma unroll
      for (int l = 0; l < GRP; ++l) {
        const uint32_t i = inc[min(q0 + l, q0 + k - 1)], par16 = (l & 1) * 16;
        if (l < k) w[l] = ((i & 0x3FF) * REC + par16) | ((((i >> 10) & 0x3FF) * REC + par16) << 16);
      }
      wv = (uint32_t)(v * REC);
      r2k = k < 32 ? r2ktab[k] : 1.0 / (2.0 * k);
      thr = k <= GRP ? 4.0e-31f : __builtin_inff();   // degree > 8: always the exact path
      // dependencies inside a sweep: the lower-numbered interior neighbours (each is seen in two cells: counted twice);
      // per cell the interior ranks of the higher-numbered ones (the vertices this one releases when it is scheduled)
      int dep = 0;
      uint32_t hw[GRP];
#pragma unroll
      for (int l = 0; l < GRP; ++l) hw[l] = 0xFFFFFFFFu;
      for (int i = q0; i < q0 + k; ++i) {
        const uint32_t wi = inc[i];
        const int a = wi & 0x3FF, c = (wi >> 10) & 0x3FF;
        const uint32_t ra = rk[a], rc = rk[c];
        dep += (a < v && ra != 0xFFFF) + (c < v && rc != 0xFFFF);
        if (i - q0 < GRP) hw[i - q0] = (a > v ? ra : 0xFFFFu) | ((c > v ? rc : 0xFFFFu) << 16);
      }
      indeg2[r] = dep;
      if (dep == 0) atomicOr(&rb[r >> 6], 1ull << (r & 63));
      if (k > GRP) atomicOr(&bigdeg[r >> 6], 1ull << (r & 63));
#pragma unroll
      for (int l = 0; l < GRP; l += 4) *reinterpret_cast<u4*>(hwt + (r * GRP + l) * 4) = u4{hw[l], hw[l + 1], hw[l + 2], hw[l + 3]};
    } else {
      w[0] = (uint32_t)AREC | ((uint32_t)CREC << 16);
    }
#pragma unroll
    for (int l = 0; l < GRP; l += 4) *reinterpret_cast<u4*>(rows + r * ROW + 4 * l) = u4{w[l], w[l + 1], w[l + 2], w[l + 3]};
    *reinterpret_cast<u4*>(rows + r * ROW + 32) = u4{(uint32_t)__double2loint(r2k), (uint32_t)__double2hiint(r2k), wv, __float_as_uint(thr)};
  }
  MDQ_SMOOTH_PHASE()   /* 5: rows */
  __syncthreads();
  // passes of one sweep by LIST SCHEDULING (wave 0): a vertex is ready when its lower-numbered interior neighbours are
  // scheduled; every pass takes the (up to) 8 ready vertices with the smallest indices, then releases their
  // higher-numbered neighbours.  Lowest-index-first follows the long ascending chains of the mesh numbering: 119
  // passes for ys930 (level by level: 141; the critical path: 113), and no level computation at all.  The ready set
  // is a bitmap over the interior ranks (one 64-bit word per lane 0..15), the release runs 8 lanes per scheduled
  // vertex (one incident cell each) with LDS atomics.
  int npass_w = 0;
  if (tid < 64) {
    const int lane = tid, l = lane & 7, g = lane >> 3;
    int remaining = n_int, p = 0, nbp = 0, lastbp = -1;
    while (remaining > 0) {
      unsigned long long word = lane < SNV / 64 ? rb[lane] : 0ull;
      const int c = __popcll(word);
      // inclusive prefix sums over the 16 bitmap lanes (one DPP row): row_shr 1, 2, 4, 8
      int incl = c;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
      const int total = __builtin_amdgcn_readlane(incl, SNV / 64 - 1);
      int take = min(total, GRP);
      const int excl = incl - c;
      if (take == 0) {
        // cannot happen on a consistent mesh (the smallest unscheduled vertex is always ready); stay safe anyway:
        // schedule the smallest unscheduled rank alone
        int first = n_int;
        for (int r = lane; r < n_int; r += 64)
          if (indeg2[r] >= 0) first = min(first, r);
        for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
        if (lane == 0) passtab[p * GRP] = (uint16_t)(first * ROW);
        take = 1;
      } else if (lane < SNV / 64) {
        // (a walk over the words with one lane per BIT and ballots was tried: slower, 234 k cycles against 167 k)
        const int mine = min(max(GRP - excl, 0), c);     // how many of this lane's lowest ready bits are taken
        for (int i = 0; i < mine; ++i) {
          const int bit = __ffsll((long long)word) - 1;
          passtab[p * GRP + excl + i] = (uint16_t)((lane * 64 + bit) * ROW);
          word &= word - 1;
        }
        if (mine > 0) rb[lane] = word;
      }
      if (lane >= take && lane < GRP) passtab[p * GRP + lane] = (uint16_t)(n_int * ROW);   // empty slots
      // release: group g = scheduled vertex g, lane l = its l-th cell
      bool big = false;
      if (g < take) {
        const int r = passtab[p * GRP + g] / ROW;
        const uint32_t h2 = *reinterpret_cast<const uint32_t*>(hwt + (r * GRP + l) * 4);
        big = (bigdeg[r >> 6] >> (r & 63)) & 1ull;
        if (l == 0) indeg2[r] = -1;                 // scheduled
        auto release = [&](uint32_t ru) {
          if (ru != 0xFFFF && atomicSub(&indeg2[ru], 1) == 1) atomicOr(&rb[ru >> 6], 1ull << (ru & 63));
        };
        release(h2 & 0xFFFF);
        release(h2 >> 16);
        if (big && l == 0) {                          // degree > 8: the cells beyond the eighth
          const int v = ivert[r], q0 = ptr[v], k = ptr[v + 1] - q0;
          for (int i = q0 + GRP; i < q0 + k; ++i) {
            const uint32_t wi = inc[i];
            const int a = wi & 0x3FF, cc = (wi >> 10) & 0x3FF;
            if (a > v) release(rk[a]);
            if (cc > v) release(rk[cc]);
          }
        }
      }
      // pass pairs (the walk is unrolled by two) that hold a vertex of more than 8 cells: the speculative sweeps run
      // them as careful passes
      if (__builtin_amdgcn_ballot_w64(big) && lastbp != (p & ~1)) {
        lastbp = p & ~1;
        if (lane == 0 && nbp < MAXBP) part[2 + nbp] = lastbp;
        ++nbp;
      }
      remaining -= take;
      ++p;
    }
    if (lane == 0) part[1] = nbp;
    // an even number of passes (the walk below is unrolled by two) and the first three passes again behind the last
    // one (the prefetch runs up to three passes ahead, into the next sweep)
    if (p & 1) {
      if (lane < GRP) passtab[p * GRP + lane] = (uint16_t)(n_int * ROW);
      ++p;
    }
    if (lane < 3 * GRP) passtab[p * GRP + lane] = passtab[lane];
    npass_w = p;
  }
  if (tid == 0) part[0] = npass_w;
  __syncthreads();
  const int npass = part[0], nbp = part[1];
  // vertex records: (x, y) for the even lanes, (y, x) for the odd lanes; record SNV: zeros, SNV + 1 / + 2: the far edge
  for (int v = tid; v < SNV + 3; v += SWG) {
    d2 p = {0.0, 0.0};
    if (v < nv) {
      const double2 xv = x[v];
      p = d2{xv.x, xv.y};
    } else if (v > SNV) {
      p = d2{(double)(v - SNV - 1), -1000.0};
    }
    *reinterpret_cast<d2*>(recb + v * REC) = p;
    *reinterpret_cast<d2*>(recb + v * REC + 16) = d2{p.y, p.x};
  }
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 7: pass table */
  // ---------------- the sweeps: wave 0 walks the passes
  if (tid < 64 && npass > 0) {
    const int lane = tid, l = lane & 7;
    const uint32_t par16 = (lane & 1) * 16;
    const int gsh = lane & ~7;
    const uint32_t lrow = OFF_ROW + 4 * l;                 // this lane's word of a metadata row
    const uint32_t st1 = 8 * (lane & 1), st2 = 24 - 8 * (lane & 1);   // where a lane's component goes inside a record
    const lds_u8* R = (const lds_u8*)lds;
    lds_u8* RW = (lds_u8*)lds;
    // pass p: the lane's cell words W (record addresses a | c << 16) and M = (1 / 2k, vertex record, lower limit) of the
    // group's vertex; RKN: row offsets of pass p + 1 (its metadata are fetched into W1 / M1 during pass p), RK2: those
    // of pass p + 2 (fetched from the pass table at PT)
#define MDQ_SMOOTH_PASS(W, M, RKN, W1, M1, RK2, PT)                                                                     \
  {                                                                                                                     \
    d2 pa, pc, pv;                                                                                                      \
    const uint32_t vrec = (M).z;                                                                                        \
    /* positions first, then the prefetches; wait for the positions only */                                            \
    asm volatile(                                                                                                       \
        "ds_read_b128 %0, %6\n\t"                                                                                       \
        "ds_read_b128 %1, %7\n\t"                                                                                       \
        "ds_read_b128 %2, %8\n\t"                                                                                       \
        "ds_read_b32 %3, %9\n\t"                                                                                        \
        "ds_read_b128 %4, %10 offset:%c12\n\t"                                                                          \
        "ds_read_u16 %5, %11\n\t"                                                                                       \
        "s_waitcnt lgkmcnt(3)"                                                                                          \
        : "=&v"(pa), "=&v"(pc), "=&v"(pv), "=&v"(W1), "=&v"(M1), "=&v"(RK2)                                             \
        : "v"((W) & 0xFFFF), "v"((W) >> 16), "v"(vrec | par16), "v"((RKN) + lrow), "v"(RKN), "v"(PT), "n"(OFF_ROW + 32) \
        : "memory");                                                                                                    \
    MDQ_SMOOTH_STAMP(t_ready)                                                                                           \
    /* component 0 of a lane is the one it keeps (even lanes x, odd lanes y), component 1 the one it hands over */      \
    double S = pa.x + pc.x;                                                                                             \
    const double T = pa.y + pc.y;                                                                                       \
    const double e0 = pc.x - pa.x, e1 = pc.y - pa.y;                                                                    \
    const double w0 = pv.x - pa.x, w1_ = pv.y - pa.y;                                                                   \
    const double cr = __builtin_fma(e1, w0, -(e0 * w1_));                                                               \
    const double len2 = __builtin_fma(e0, e0, e1 * e1);                                                                 \
    /* squared altitude, ~1e-6 relative; an unused lane (both vertices the zero record) gives 0 * inf = NaN, whose */   \
    /* bit pattern sorts above every number: never the minimum (three DPP-fused unsigned minima over the 8 lanes) */    \
    uint32_t ab = __float_as_uint((float)(cr * cr) * __builtin_amdgcn_rcpf((float)len2));                               \
    /* centroid sums: stage 1 exchanges the other component with the neighbour lane, stages 2 and 3 add like */         \
    /* components (lanes 0..3 of the group end with the totals: lane 0 x, lane 1 y) */                                  \
    S += dpp8<0xB1>(T);                                                                                                 \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(ab));               \
    S += dpp8<0x4E>(S);                                                                                                 \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(ab));               \
    {                                                                                                                   \
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(S), 0x104, 0xF, 0x5, true); /* row_shl:4 */          \
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(S), 0x104, 0xF, 0x5, true);                          \
      S += __hiloint2double(hi, lo);                                                                                    \
    }                                                                                                                   \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(ab));                         \
    const double d = __builtin_fma(S, __hiloint2double((int)(M).y, (int)(M).x), -pv.x); /* lane 0: dx, lane 1: dy */    \
    const float df = (float)d, t2 = df * df;                                                                            \
    float q2f;                                                                                                          \
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(q2f) : "v"(t2));    \
    /* fast decision (lanes 0 and 1 hold the same q2f and minimum): the vertex moves (|d| clearly above DOLFIN_EPS) */  \
    /* and the full step is clearly allowed (q2 < r_min^2 / 4 with a 1e-3 margin over the fp32 errors) */               \
    const bool fast = (q2f > __uint_as_float((M).w)) & (q2f < 0.24975f * __uint_as_float(ab));                          \
    double pn = pv.x + d;                                                                                               \
    const unsigned long long slowm = 0x0101010101010101ull & ~__builtin_amdgcn_ballot_w64(fast);                        \
    if (slowm) { /* wave-uniform and rare: exact fp64 update for the groups that could not decide */                    \
      if ((slowm >> gsh) & 1ull)                                                                                        \
        exact_update(R, (const lds_i32*)(lds + OFF_PTR), (const lds_u32*)(lds + OFF_INC), (int)(vrec / REC), l, pn);    \
    }                                                                                                                   \
    /* the prefetched metadata has long arrived */                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W1), "+v"(M1), "+v"(RK2)::"memory");                                     \
    if (l < 2) asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %2" ::"v"(vrec + st1), "v"(vrec + st2), "v"(pn) : "memory"); \
    MDQ_SMOOTH_TRACE_OUT(vrec)                                                                                          \
  }
#ifdef MDQ_SMOOTH_DIAGNOSE   /* debug builds: the last undecided update of a speculative sweep in slots 40.. */
#define MDQ_SMOOTH_DIAG(fast, vrec, q2f, ab, thr)                                                     \
  if (l == 0 && !(fast) && (vrec) != (uint32_t)ZREC) {                                                 \
    g_abandoned[40] = b; g_abandoned[41] = (vrec) / REC; g_abandoned[42] = __float_as_uint(q2f);       \
    g_abandoned[43] = ab; g_abandoned[44] = thr; g_abandoned[45] = sweep; g_abandoned[46] = nv;        \
  }
#else
#define MDQ_SMOOTH_DIAG(fast, vrec, q2f, ab, thr)
#endif
#ifdef MDQ_SMOOTH_TRACE
#define MDQ_SMOOTH_STAMP(t) const long long t = clock64();
#define MDQ_SMOOTH_TRACE_OUT(vrec)                                     \
  if (trace && b == 0 && l == 0 && (vrec) != (uint32_t)ZREC) {         \
    trace[2 * ((int64_t)sweep * SNV + (vrec) / REC)] = t_ready;        \
    trace[2 * ((int64_t)sweep * SNV + (vrec) / REC) + 1] = clock64();  \
  }
#else
#define MDQ_SMOOTH_STAMP(t)
#define MDQ_SMOOTH_TRACE_OUT(vrec)
#endif
    // ---- speculative walk: the new position is stored BEFORE the step-limit test of the update has been evaluated;
    // the test (same fp32 decision, from the registers the update was computed from) runs behind the position loads of
    // the next pass, i.e. in their LDS latency.  An update that is not clearly a full step (none on the reference
    // meshes) abandons the walk: the records are reloaded and the sweeps run again with the test in front of the store.
#define MDQ_SMOOTH_SPEC(PA, PC, PV, M, PA1, PC1, PV1, W1, M1, RKIN, RKOUT, PT)                                          \
  {                                                                                                                     \
    MDQ_SMOOTH_STAMP(t_ready)                                                                                           \
    double S = PA.x + PC.x;                                                                                             \
    const double T = PA.y + PC.y;                                                                                       \
    S += dpp8<0xB1>(T);                                                                                                 \
    S += dpp8<0x4E>(S);                                                                                                 \
    {                                                                                                                   \
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(S), 0x104, 0xF, 0x5, true); /* row_shl:4 */          \
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(S), 0x104, 0xF, 0x5, true);                          \
      S += __hiloint2double(hi, lo);                                                                                    \
    }                                                                                                                   \
    const double d = __builtin_fma(S, __hiloint2double((int)(M).y, (int)(M).x), -PV.x); /* lane 0: dx, lane 1: dy */    \
    const double pn = PV.x + d;                                                                                         \
    const uint32_t vrec = (M).z, thr = (M).w;                                                                           \
    if (l < 2) {                                                                                                        \
      *reinterpret_cast<lds_f64*>(RW + vrec + st1) = pn;                                                                \
      *reinterpret_cast<lds_f64*>(RW + vrec + st2) = pn;                                                                \
    }                                                                                                                   \
    MDQ_SMOOTH_TRACE_OUT(vrec)                                                                                          \
    asm volatile("" ::: "memory");                                                                                      \
    /* positions of the next pass, metadata of the one behind it, vertex list of the third */                           \
    PA1 = *reinterpret_cast<const lds_d2*>(R + ((W1) & 0xFFFF));                                                        \
    PC1 = *reinterpret_cast<const lds_d2*>(R + ((W1) >> 16));                                                           \
    PV1 = *reinterpret_cast<const lds_d2*>(R + ((M1).z | par16));                                                       \
    const uint32_t wnew = *reinterpret_cast<const lds_u32*>(R + (RKIN) + lrow);                                         \
    const u4 mnew = *reinterpret_cast<const lds_u4*>(R + (RKIN) + (OFF_ROW + 32));                                      \
    RKOUT = *reinterpret_cast<const lds_u16*>(R + (PT));                                                                \
    /* the step-limit test of THIS pass */                                                                              \
    const double e0 = PC.x - PA.x, e1 = PC.y - PA.y;                                                                    \
    const double w0 = PV.x - PA.x, w1_ = PV.y - PA.y;                                                                   \
    const double cr = __builtin_fma(e1, w0, -(e0 * w1_));                                                               \
    const double len2 = __builtin_fma(e0, e0, e1 * e1);                                                                 \
    uint32_t ab = __float_as_uint((float)(cr * cr) * __builtin_amdgcn_rcpf((float)len2));                               \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(ab));               \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(ab));               \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(ab));                         \
    const float df = (float)d, t2 = df * df;                                                                            \
    float q2f;                                                                                                          \
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(q2f) : "v"(t2));    \
    const bool fast = (q2f > __uint_as_float(thr)) & (q2f < 0.24975f * __uint_as_float(ab));                            \
    undecided |= 0x0101010101010101ull & ~__builtin_amdgcn_ballot_w64(fast);                                            \
    MDQ_SMOOTH_DIAG(fast, vrec, q2f, ab, thr)                                                                          \
    W1 = wnew;                                                                                                          \
    M = mnew;                                                                                                           \
  }
    // The walk alternates between the two kinds of sweeps.  A vertex removal leaves a cavity whose neighbours take a
    // LIMITED first step in most environments (tools: 5 of 8 removals on ys930, always in sweep 0), so the first sweeps
    // run careful; the rest run speculative from a checkpoint of the records in the coordinate array (this kernel owns
    // it until the write-back).  A speculative sweep with an undecided update is abandoned: back to the checkpoint,
    // careful through that sweep and `extra` more (doubling), checkpoint, speculative again.
    const uint32_t lpt = OFF_PT + 2 * (lane >> 3);         // this group's slot of a pass
    int done = 0, ncareful = iters < 3 || nbp > MAXBP ? iters : 3, extra = 2;
    while (true) {
      if (ncareful > 0) {
        // ---- careful sweeps: decision in front of every store, exact fp64 path where needed
        uint32_t wA, wB = 0, rkA = 0, rkB;
        u4 mA, mB = {0, 0, 0, 0};
        {
          const uint32_t rk0 = passtab[lane >> 3];
          wA = *reinterpret_cast<const uint32_t*>(rows + rk0 + 4 * l);
          mA = *reinterpret_cast<const u4*>(rows + rk0 + 32);
          rkB = passtab[GRP + (lane >> 3)];
        }
        for (int sweep = done; sweep < done + ncareful; ++sweep) {
          uint32_t pt = lpt + 2 * (GRP * 2);
          for (int q = 0; q < npass; q += 2) {
            MDQ_SMOOTH_PASS(wA, mA, rkB, wB, mB, rkA, pt)
            MDQ_SMOOTH_PASS(wB, mB, rkA, wA, mA, rkB, pt + GRP * 2)
            pt += 2 * (GRP * 2);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        done += ncareful;
      }
      if (done >= iters) break;
      // ---- checkpoint, then speculative sweeps
      for (int v = lane; v < nv; v += 64) {
        const d2 p = *reinterpret_cast<const d2*>(recb + v * REC);
        x[v] = double2{p.x, p.y};
      }
      int bad = -1;
      {
        unsigned long long undecided = 0;   // groups whose update was not clearly a full step (tested once per sweep: a
                                            // branch behind every test would keep the next loads behind it)
        // pass 0 in set A, pass 1 in set B, vertex list of pass 2
        d2 paA, pcA, pvA, paB = {0, 0}, pcB = {0, 0}, pvB = {0, 0};
        uint32_t wN, rkP, rkQ = 0;     // wN: cell words of the pass whose positions are loaded next
        u4 mA, mB;
        {
          const uint32_t rk0 = passtab[lane >> 3], rk1 = passtab[GRP + (lane >> 3)];
          const uint32_t wA = *reinterpret_cast<const uint32_t*>(rows + rk0 + 4 * l);
          mA = *reinterpret_cast<const u4*>(rows + rk0 + 32);
          wN = *reinterpret_cast<const uint32_t*>(rows + rk1 + 4 * l);
          mB = *reinterpret_cast<const u4*>(rows + rk1 + 32);
          rkP = passtab[2 * GRP + (lane >> 3)];
          paA = *reinterpret_cast<const lds_d2*>(R + (wA & 0xFFFF));
          pcA = *reinterpret_cast<const lds_d2*>(R + (wA >> 16));
          pvA = *reinterpret_cast<const lds_d2*>(R + (mA.z | par16));
        }
        if (nbp == 0) {
          for (int sweep = done; sweep < iters; ++sweep) {
            uint32_t pt = lpt + 3 * (GRP * 2);
            for (int q = 0; q < npass; q += 2) {
              // pass q (set A): loads the positions of pass q + 1 (set B), the metadata of pass q + 2 (into set A)
              MDQ_SMOOTH_SPEC(paA, pcA, pvA, mA, paB, pcB, pvB, wN, mB, rkP, rkQ, pt)
              MDQ_SMOOTH_SPEC(paB, pcB, pvB, mB, paA, pcA, pvA, wN, mA, rkQ, rkP, pt + GRP * 2)
              pt += 2 * (GRP * 2);
            }
            if (undecided) {
              bad = sweep;
              break;
            }
          }
        } else {
          // the mesh has vertices of more than 8 cells (the fast path has one lane per cell): the listed pass pairs run
          // careful (exact path for those vertices), the passes between them speculative as above; both pipelines
          // are primed again at every switch (~1.5 % of a sweep per listed pair)
          for (int sweep = done; sweep < iters; ++sweep) {
            int q0 = 0;
            for (int k = 0; k <= nbp; ++k) {
              const int bp = k < nbp ? part[2 + k] : npass;
              if (bp > q0) {
                {
                  const uint32_t rk0 = passtab[q0 * GRP + (lane >> 3)], rk1 = passtab[(q0 + 1) * GRP + (lane >> 3)];
                  const uint32_t wA = *reinterpret_cast<const uint32_t*>(rows + rk0 + 4 * l);
                  mA = *reinterpret_cast<const u4*>(rows + rk0 + 32);
                  wN = *reinterpret_cast<const uint32_t*>(rows + rk1 + 4 * l);
                  mB = *reinterpret_cast<const u4*>(rows + rk1 + 32);
                  rkP = passtab[(q0 + 2) * GRP + (lane >> 3)];
                  paA = *reinterpret_cast<const lds_d2*>(R + (wA & 0xFFFF));
                  pcA = *reinterpret_cast<const lds_d2*>(R + (wA >> 16));
                  pvA = *reinterpret_cast<const lds_d2*>(R + (mA.z | par16));
                }
                uint32_t pt = lpt + (q0 + 3) * (GRP * 2);
                for (int q = q0; q < bp; q += 2) {
                  MDQ_SMOOTH_SPEC(paA, pcA, pvA, mA, paB, pcB, pvB, wN, mB, rkP, rkQ, pt)
                  MDQ_SMOOTH_SPEC(paB, pcB, pvB, mB, paA, pcA, pvA, wN, mA, rkQ, rkP, pt + GRP * 2)
                  pt += 2 * (GRP * 2);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              }
              if (bp < npass) {
                uint32_t cwA, cwB = 0, crkA = 0, crkB;
                u4 cmA, cmB = {0, 0, 0, 0};
                {
                  const uint32_t rk0 = passtab[bp * GRP + (lane >> 3)];
                  cwA = *reinterpret_cast<const uint32_t*>(rows + rk0 + 4 * l);
                  cmA = *reinterpret_cast<const u4*>(rows + rk0 + 32);
                  crkB = passtab[(bp + 1) * GRP + (lane >> 3)];
                }
                const uint32_t pt = lpt + (bp + 2) * (GRP * 2);
                MDQ_SMOOTH_PASS(cwA, cmA, crkB, cwB, cmB, crkA, pt)
                MDQ_SMOOTH_PASS(cwB, cmB, crkA, cwA, cmA, crkB, pt + GRP * 2)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              }
              q0 = bp + 2;
            }
            if (undecided) {
              bad = sweep;
              break;
            }
          }
        }
      }
      if (bad < 0) break;
      if (lane == 0) atomicAdd(&g_abandoned[bad < 63 ? bad : 62], 1ull);
      // ---- back to the checkpoint (each lane reads what it wrote itself)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      for (int v = lane; v < nv; v += 64) {
        const double2 xv = x[v];
        *reinterpret_cast<d2*>(recb + v * REC) = d2{xv.x, xv.y};
        *reinterpret_cast<d2*>(recb + v * REC + 16) = d2{xv.y, xv.x};
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      ncareful = bad - done + 1 + extra;
      if (ncareful > iters - done) ncareful = iters - done;
      extra *= 2;
    }
#undef MDQ_SMOOTH_SPEC
#undef MDQ_SMOOTH_PASS
#undef MDQ_SMOOTH_STAMP
#undef MDQ_SMOOTH_TRACE_OUT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  MDQ_SMOOTH_PHASE()   /* 8: sweeps */
  for (int v = tid; v < nv; v += SWG) {
    const d2 p = *reinterpret_cast<const d2*>(recb + v * REC);
    x[v] = double2{p.x, p.y};
  }
  MDQ_SMOOTH_PHASE()   /* 9: written back */
#undef MDQ_SMOOTH_PHASE
}

// B environments over the workgroups of the grid (mdq_smooth: one workgroup per environment; the hand-back launch of
// mdq_smooth_fast: ONE workgroup that walks over the - normally zero - environments with iterations left: a single
// workgroup of 141 KB LDS finds a compute unit at once, 128 of them wait for the kernels of other streams to drain)
__global__ __launch_bounds__(SWG) void smooth_kernel(int B, int NV, int NT, double* coords, const int32_t* cells,
                                                     const int32_t* nv_, const int32_t* nt_, const int32_t* iters_,
                                                     int cap, long long* trace) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  if (gridDim.x < (unsigned)B) {
    // the walk-over form: is there anything to do at all?  Every wave looks at all the counts (two loads per lane for 128
    // environments, no LDS, no barrier): stepping through 128 early returns one by one cost 48 us of dependent loads
    bool any = false;
    for (int b = threadIdx.x & 63; b < B; b += 64) any |= iters_[b] > 0;
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
  }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    smooth_env(lds, b, NV, NT, coords, cells, nv_, nt_, iters_, cap, trace);
    __syncthreads();
  }
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

extern "C" int mdq_smooth_stats(int64_t* out64, int32_t reset) {
  if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(mdq_smoothing::g_abandoned), 64 * sizeof(int64_t)) != hipSuccess)
    return mdq_set_error("mdq_smooth_stats: copy failed");
  if (reset) {
    const int64_t zero[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_smoothing::g_abandoned), zero, sizeof(zero)) != hipSuccess)
      return mdq_set_error("mdq_smooth_stats: reset failed");
  }
  return 0;
}

extern "C" int64_t mdq_smooth_workspace_bytes(int32_t B, int32_t NV, int32_t NT) {
  if (B <= 0 || NV <= 0 || NT <= 0) return 0;
  if (NV <= mdq_smoothing::SNV && NT <= mdq_smoothing::SNT) return 0;          // every table in LDS
  return smooth_big_workspace_bytes(B, NV, NT);
}

extern "C" int mdq_smooth(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                          const int32_t* nt, const int32_t* iterations, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B <= 0 || !coords || !cells || !nv || !nt || !iterations) return mdq_set_error("mdq_smooth: bad arguments");
  if (NV > mdq_smoothing::SNV || NT > mdq_smoothing::SNT)     // a mesh beyond the 1024-vertex kernels: level-scheduled kernel
    return smooth_big_launch(B, NV, NT, coords, cells, nv, nt, iterations, nullptr, nullptr, 0, workspace, workspace_bytes, stream);
  long long* trace = nullptr;
#ifdef MDQ_SMOOTH_TRACE
  trace = mdq_smooth_trace_host();
#endif
  hipLaunchKernelGGL(mdq_smoothing::smooth_kernel, dim3(B), dim3(mdq_smoothing::SWG), 0, (hipStream_t)stream, B, NV, NT, coords,
                     cells, nv, nt, iterations, 0, trace);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("smooth_kernel launch failed");
  return 0;
}
