// DOLFIN `Mesh.smooth(n)` (flow_solver.py:65-67, 236-237) - the FULL-STEP sweeps as a blocked triangular solve.
//
// A Gauss-Seidel sweep in which every interior vertex takes the full step to the centroid of its neighbours is LINEAR,
// with a matrix that depends on the topology only:
//      (D - L) x_new = U x_old + (boundary terms),     D = number of neighbours, L / U = adjacency towards the lower /
//                                                      higher numbered interior vertices,
// and the x and y components are two independent right-hand sides.  The per-vertex walk of mdq_smooth.hip pays one
// dependent pass per 8 updates (119 passes per sweep on ys930, ~410 cycles each: the mesh numbering makes a sweep a chain
// of 113 levels).  Here the interior ranks are cut into blocks of 32 consecutive ranks and
//      x_B = M_B g_B,     M_B = (I - D_B^-1 L_BB)^-1 D_B^-1  (32 x 32, lower triangular, built ONCE per launch),
//                         g_B = sum of the neighbour positions that are not lower-numbered members of the same block
//                               (read from the in-place position array: new values for lower blocks, old ones otherwise),
// so that a sweep is 22 dependent block steps (gather 7 positions per lane -> 32 x 32 mat-vec, 16 FMAs per lane) instead
// of 119 passes (tools/chain_depth.py, tools/smooth_block_proto.py: contiguous index blocks form a pure chain, so there
// is nothing to gain from more waves on the solve itself).  One wave per component solves; the rows of M_B a lane needs
// stream from L2 (176 KB per mesh: they do not fit the LDS next to the positions) two blocks ahead.
//
// Whether every update of a sweep really WAS a clear full step (DOLFIN limits a step to half the smallest altitude of
// the incident cells, and leaves a vertex alone that is closer than DOLFIN_EPS to its centroid) is checked afterwards and
// in parallel: sweep s is validated by the other waves WHILE the solver waves run sweep s + 1, from two snapshots of the
// position array (before / after sweep s: the positions an update saw are the new ones of its lower-numbered interior
// neighbours and the old ones of everything else), with the same conservative fp32 decision as mdq_smooth.hip.  A sweep
// with an update that is not clearly a full step is rolled back to its snapshot and redone (below): exact sequential
// semantics either way.  Results are bitwise reproducible (fixed summation orders) and agree with the sequential loop to
// round-off (different association: 1e-15 per sweep, 4e-15 after 47 sweeps on the lab meshes).
//
// Limited steps (right after a vertex removal the cavity's neighbours take them in the first sweep or two) are handled here
// as well: a sweep that fails its validation is redone from its snapshot with the offending vertices FLAGGED - they get
// DOLFIN's exact update (exact_vertex), and the rows of their block behind them the correction M[:, i] (x* - x_i) / M_ii,
// i.e. the linear solve with row i's right-hand side moved so that x_i = x* (repair_sweep) - until a round flags nothing
// new.  The kernel starts in this CHECKED mode (solve, validate with all waves, repair) and switches to the PIPELINED mode
// above after the first sweep that passes at the first try.
//
// mdq_smooth_fast (below) = this kernel -> mdq_smooth's kernel for the environments that were handed back (a mesh beyond
// this kernel's limits - more than 16 cells at a vertex, 14 gather slots - or a sweep that did not settle within MAXROUNDS
// repair rounds; normally none: it returns at once).
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_smooth_lin {
constexpr int LNV = 1024;          // vertex capacity
constexpr int LNT = 2048;          // triangle capacity
#ifndef MDQ_SMOOTH_LWG
#define MDQ_SMOOTH_LWG 768
#endif
constexpr int LWG = MDQ_SMOOTH_LWG;   // threads per workgroup (512 or 768): waves 0 / 1 solve x / y, the waves on the other two SIMDs validate
constexpr int SCT = 512;           // threads that take part in the block scans (two entries each)
constexpr int TRW = LWG / 64 < 11 ? LWG / 64 : 11;   // waves that build block inverses (their packed triangles share 3 position buffers)
static_assert(LWG == 512 || LWG == 768, "workgroup shape");
constexpr int BS = 32;             // rows per block
constexpr int NSLOT = 14;          // gather slots per row (7 per lane half)
constexpr int MAXNB = 16;          // neighbours / cells per vertex the setup handles (more: the mesh goes to the careful walk)
constexpr int MAXLOW = 8;          // lower-numbered neighbours inside the own block
constexpr int SROW = 32;           // bytes per solver row: [7 x u16 gather offsets, u16 store offset] x 2 halves
constexpr int ZOFF = LNV * 16;     // the zero record of a position buffer (byte offset)
constexpr int PBUF = (LNV + 1) * 16;
constexpr int MBLK = 2 * 8 * 32 * 2;   // doubles per block of M in the workspace: [half][t][row][2]
constexpr int MAXROUNDS = 5;       // repair rounds of one checked sweep before the careful walk takes over

constexpr int OFF_CUR = 0;                                   // positions, updated in place by the solver waves
constexpr int OFF_SNAP = OFF_CUR + PBUF;                     // two snapshots (state at the start of sweep s: slot s & 1)
constexpr int OFF_SROW = OFF_SNAP + 2 * PBUF;                // solver rows
constexpr int OFF_PTR = OFF_SROW + LNV * SROW;               // vertex -> cells
constexpr int OFF_INC = OFF_PTR + ((LNV + 1) * 4 + 12);      // per (vertex, cell), ascending cell: a | c << 10 | v << 20 | new(a) << 30 | new(c) << 31
constexpr int OFF_RK = OFF_INC + 3 * LNT * 4;                // interior rank (0xFFFF: not interior)
constexpr int OFF_IVERT = OFF_RK + LNV * 2;                  // interior vertices in index order
constexpr int OFF_LMETA = OFF_IVERT + LNV * 2;               // per rank: local indices of the lower neighbours inside the block
constexpr int OFF_KDEG = OFF_LMETA + LNV * MAXLOW;           // per rank: number of neighbours | in-block lower count << 8
constexpr int OFF_G = OFF_KDEG + LNV * 2;                    // g of the block in flight, per component
constexpr int OFF_R2K = OFF_G + 2 * BS * 8;                  // 1 / (2 k)
constexpr int OFF_MISC = OFF_R2K + 32 * 8;                   // [0] n_int  [1] bad / newly flagged  [2] eligible
constexpr int OFF_FIXV = OFF_MISC + 64;                      // per vertex: takes the exact update in the sweep at hand
constexpr int OFF_PART = OFF_FIXV + LNV + 16;
constexpr int LDS_BYTES = OFF_PART + LWG * 4;
// setup scratch inside the position buffers (the positions are loaded last): per-vertex neighbour lists, then the rows of
// the block inverses under construction
constexpr int OFF_NB = OFF_CUR;                              // [LNV][MAXNB] u16: id | count << 10
constexpr int OFF_TRI = OFF_CUR;                             // 8 waves x 528 doubles (packed lower triangles)
constexpr int OFF_TMP = OFF_SROW;                            // cell lists in arrival order (before the rows are built)
static_assert(LNV * MAXNB * 2 <= 3 * PBUF && TRW * 528 * 8 <= 3 * PBUF && 3 * LNT * 4 <= LNV * SROW, "setup scratch");
static_assert(OFF_SROW % 16 == 0 && OFF_PTR % 16 == 0 && OFF_INC % 16 == 0 && OFF_G % 16 == 0 && OFF_R2K % 8 == 0, "LDS alignment");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over groups of 8 consecutive lanes (all lanes get the total): xor 1, xor 2 (quad_perm), other quad (row_half_mirror)
__device__ __forceinline__ double grp8_sum(double v) {
  v += dppd<0xB1>(v);
  v += dppd<0x4E>(v);
  v += dppd<0x141>(v);
  return v;
}
__device__ __forceinline__ float grp8_min(float v) {
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  return v;
}
// v[lane & 31] + v[(lane & 31) + 32] in every lane (lower half first: the same sum in both halves)
__device__ __forceinline__ double halves_sum(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

__device__ __forceinline__ void scan_inclusive(int* data, int* part) {   // data[0..LNV): the first SCT threads, 2 entries each
  const int tid = threadIdx.x;
  constexpr int PER = LNV / SCT;
  const bool sc = tid < SCT;
  int loc[PER], run = 0;
  if (sc) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      run += data[tid * PER + i];
      loc[i] = run;
    }
    part[tid] = run;
  }
  __syncthreads();
  for (int off = 1; off < SCT; off <<= 1) {
    const int add = (sc && tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    if (sc) part[tid] += add;
    __syncthreads();
  }
  if (sc) {
    const int base = part[tid] - run;
#pragma unroll
    for (int i = 0; i < PER; ++i) data[tid * PER + i] = base + loc[i];
  }
  __syncthreads();
}

// one validation pass: lane = one (vertex, incident cell) entry of the flat list.  Did the update its vertex took in the
// sweep between the snapshots OLD and NEW clearly pass DOLFIN's two tests?  The step d = NEW[v] - OLD[v] IS the step to
// the centroid (that is what the solve computed); it must be clearly longer than DOLFIN_EPS and clearly shorter than half
// the altitude over this cell's opposite edge, taken with the positions the update saw (the new ones of lower-numbered
// interior neighbours, the old ones otherwise) - the same conservative fp32 decision as mdq_smooth.hip, per cell instead
// of on the minimum over the cells.  Returns true if this entry is NOT clearly a full step.
template <bool FLAG>
__device__ __forceinline__ bool validate_entry(unsigned char* lds, int e, int ne, const unsigned char* OLD,
                                               const unsigned char* NEW) {
  const uint32_t* inc = reinterpret_cast<const uint32_t*>(lds + OFF_INC);
  const uint32_t w = e < ne ? inc[e] : 0xFFFFFFFFu;
  const int a = w & 0x3FF, c = (w >> 10) & 0x3FF, v = (w >> 20) & 0x3FF;
  const d2 pa = *reinterpret_cast<const d2*>(((w >> 30) & 1 ? NEW : OLD) + a * 16);
  const d2 pc = *reinterpret_cast<const d2*>((w >> 31 ? NEW : OLD) + c * 16);
  const d2 pv = *reinterpret_cast<const d2*>(OLD + v * 16);
  const d2 pn = *reinterpret_cast<const d2*>(NEW + v * 16);
  const float dx = (float)(pn.x - pv.x), dy = (float)(pn.y - pv.y);
  const float q2f = __builtin_fmaf(dx, dx, dy * dy);
  const double e0 = pc.x - pa.x, e1 = pc.y - pa.y, w0 = pv.x - pa.x, w1 = pv.y - pa.y;
  const double cr = __builtin_fma(e1, w0, -(e0 * w1));
  const double len2 = __builtin_fma(e0, e0, e1 * e1);
  const float alt2 = (float)(cr * cr) * __builtin_amdgcn_rcpf((float)len2);   // squared altitude, ~1e-6 relative
  const bool fast = (q2f > 4.0e-31f) & (q2f < 0.24975f * alt2);
  bool bad = (w != 0xFFFFFFFFu) & !fast;
  if (FLAG) {            // checked sweeps: vertices that already take the exact update are exempt, new ones are flagged
    unsigned char* fixv = lds + OFF_FIXV;
    bad = bad && !fixv[v];
    if (bad) fixv[v] = 1;
  }
  return bad;
}

// the block step of a solver wave (component COMP): lane = row i | half h << 5.  M: the lane's 16 entries of row i
// (columns 16 h .. 16 h + 15) as 8 pairs; meta: the lane's 8 u16 of the solver row (7 gather offsets + the store offset)
#ifdef MDQ_LIN_TRACE2
// debug build only: where the cycles of ONE block step go (wave 0 = the x component of mesh 0; s_memtime waits for the
// wave's outstanding LDS operations, so every stamp is also a fence: the sum of the parts overstates the step a little)
__device__ long long mdq_lin_bt_buf[8];
#define BT_DECL long long bt_t_ = __builtin_amdgcn_s_memtime();
#define BT(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (COMP == 0 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) mdq_lin_bt_buf[k] += tn_ - bt_t_; bt_t_ = __builtin_amdgcn_s_memtime(); }
extern "C" int mdq_lin_bt_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_lin_bt_buf), sizeof(long long) * 8) != hipSuccess) return -1;
  if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_lin_bt_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define BT_DECL
#define BT(k)
#endif

template <int COMP>
__device__ __forceinline__ void solve_block(unsigned char* lds, const u4 meta, const d2 (&M)[8], int i, int h) {
  BT_DECL
  const unsigned char* cur = lds + OFF_CUR + COMP * 8;
  const double v0 = *reinterpret_cast<const double*>(cur + (meta.x & 0xFFFF));
  const double v1 = *reinterpret_cast<const double*>(cur + (meta.x >> 16));
  const double v2 = *reinterpret_cast<const double*>(cur + (meta.y & 0xFFFF));
  const double v3 = *reinterpret_cast<const double*>(cur + (meta.y >> 16));
  const double v4 = *reinterpret_cast<const double*>(cur + (meta.z & 0xFFFF));
  const double v5 = *reinterpret_cast<const double*>(cur + (meta.z >> 16));
  const double v6 = *reinterpret_cast<const double*>(cur + (meta.w & 0xFFFF));
#ifdef MDQ_LIN_TRACE2
  double sv_ = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + v6);
  asm volatile("" : "+v"(sv_));
  BT(0)                                 // gather: 7 LDS reads + 6 adds
  const double g = halves_sum(sv_);
#else
  const double g = halves_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + v6));
#endif
  double* G = reinterpret_cast<double*>(lds + OFF_G) + COMP * BS;
  if (h == 0) G[i] = g;
  asm volatile("" ::: "memory");      // (the LDS serves a wave's operations in order: the reads below see the store)
  BT(1)                                 // cross-half add (2 permlane swaps) + store of g
  const d2* gq = reinterpret_cast<const d2*>(G + 16 * h);
  double acc0 = 0.0, acc1 = 0.0;
#ifdef MDQ_LIN_TRACE2
  d2 gg_[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) gg_[t] = gq[t];
  BT(2)                                 // the 8 reads of g (16 values) back from LDS
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    acc0 = __builtin_fma(M[t].x, gg_[t].x, acc0);
    acc1 = __builtin_fma(M[t].y, gg_[t].y, acc1);
  }
  asm volatile("" : "+v"(acc0), "+v"(acc1));
  BT(3)                                 // two chains of 8 FMAs (includes the wait for the prefetched M rows)
#else
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const d2 gg = gq[t];
    acc0 = __builtin_fma(M[t].x, gg.x, acc0);
    acc1 = __builtin_fma(M[t].y, gg.y, acc1);
  }
#endif
  const double x = halves_sum(acc0 + acc1);
  if (h == 0) *reinterpret_cast<double*>(lds + OFF_CUR + COMP * 8 + (meta.w >> 16)) = x;
  asm volatile("" ::: "memory");
  BT(4)                                 // cross-half add + store of x
}

__device__ __forceinline__ void load_m(d2 (&M)[8], const d2* mg, int b, int i, int h) {
#ifdef LIN_NOMLOAD   /* experiment: no workspace traffic (wrong results) */
#pragma unroll
  for (int t = 0; t < 8; ++t) M[t] = d2{0.01 * (b + t), 0.02 * i};
#else
  const d2* p = mg + ((size_t)(b * 2 + h) * 8) * 32 + i;
#pragma unroll
  for (int t = 0; t < 8; ++t) M[t] = p[t * 32];
#endif
}

template <int COMP>
__device__ __forceinline__ void solve_sweep(unsigned char* lds, const d2* mg, int nb, int lane) {
  const int i = lane & 31, h = lane >> 5;
  const unsigned char* srow = lds + OFF_SROW + i * SROW + h * 16;
  d2 MA[8], MB[8], MC[8];
  load_m(MA, mg, 0, i, h);
  load_m(MB, mg, 1, i, h);
  u4 meta = *reinterpret_cast<const u4*>(srow);
  for (int b = 0; b < nb; b += 3) {
    u4 mnext = *reinterpret_cast<const u4*>(srow + (b + 1) * (BS * SROW));
    load_m(MC, mg, b + 2, i, h);           // (the workspace holds two blocks of padding behind the last one)
    solve_block<COMP>(lds, meta, MA, i, h);
    if (b + 1 >= nb) break;
    meta = *reinterpret_cast<const u4*>(srow + (b + 2) * (BS * SROW));
    load_m(MA, mg, b + 3, i, h);
    solve_block<COMP>(lds, mnext, MB, i, h);
    if (b + 2 >= nb) break;
    mnext = *reinterpret_cast<const u4*>(srow + (b + 3) * (BS * SROW));
    load_m(MB, mg, b + 4, i, h);
    solve_block<COMP>(lds, meta, MC, i, h);
    meta = mnext;
  }
}

__device__ __forceinline__ double grp8_mind(double v) {
  v = fmin(v, dppd<0xB1>(v));
  v = fmin(v, dppd<0x4E>(v));
  v = fmin(v, dppd<0x141>(v));
  return v;
}
// 1/sqrt(x) and 1/x to double precision (~1 ulp): hardware estimate + two Newton steps
__device__ __forceinline__ double lin_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ double lin_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// DOLFIN's update of vertex v in exact fp64 (any degree, limited step, "stays" rule), by every group of 8 lanes of the
// wave redundantly: the vertex's old position from OLD, its neighbours' positions as the update sees them (the new ones
// of lower-numbered interior neighbours = cur, the old ones otherwise).  Same arithmetic as exact_update of mdq_smooth.hip.
__device__ __forceinline__ d2 exact_vertex(const unsigned char* lds, int v, int l, const unsigned char* OLD) {
#pragma clang fp contract(off)
  const double EPS = 3.0e-16;
  const int* ptr = reinterpret_cast<const int*>(lds + OFF_PTR);
  const uint32_t* inc = reinterpret_cast<const uint32_t*>(lds + OFF_INC);
  const unsigned char* CUR = lds + OFF_CUR;
  const int q0 = ptr[v], k = ptr[v + 1] - q0;
  const d2 p = *reinterpret_cast<const d2*>(OLD + v * 16);
  double sx = 0.0, sy = 0.0, rm = 1e300;
  for (int q = l; q < k; q += 8) {
    const uint32_t w = inc[q0 + q];
    const int a = w & 0x3FF, c = (w >> 10) & 0x3FF;
    const d2 pa = *reinterpret_cast<const d2*>(((w >> 30) & 1 ? CUR : OLD) + a * 16);
    const d2 pc = *reinterpret_cast<const d2*>((w >> 31 ? CUR : OLD) + c * 16);
    sx += pa.x + pc.x;
    sy += pa.y + pc.y;
    const double tx = pc.x - pa.x, ty = pc.y - pa.y;
    const double cr = ty * (p.x - pa.x) - tx * (p.y - pa.y);
    rm = fmin(rm, cr * cr * lin_rcp(tx * tx + ty * ty));   // SQUARED distance to the line through the opposite edge
  }
  sx = grp8_sum(sx);
  sy = grp8_sum(sy);
  rm = grp8_mind(rm);
  const double r2k = 1.0 / (2.0 * k);
  const double dx = sx * r2k - p.x, dy = sy * r2k - p.y;
  const double q2 = dx * dx + dy * dy;
  if (!(q2 >= EPS * EPS && q2 > 0.0)) return p;          // |c - p| < DOLFIN_EPS: the vertex stays
  if (0.25 * rm < q2) {                                  // limited step: needs the lengths
    const double f = 0.5 * (rm * lin_rsqrt(rm)) * lin_rsqrt(q2);
    return d2{p.x + f * dx, p.y + f * dy};
  }
  return d2{p.x + dx, p.y + dy};                         // |c - p| <= r_min / 2: to the centroid itself
}

// a sweep with vertices that do NOT take the full step (flags in OFF_FIXV), by ONE wave for both components: per block the
// same gather and mat-vec as solve_block; then, in row order, every flagged row i gets its exact update x* (its lower
// in-block neighbours are final by then) and the block's rows behind it the correction M[:, i] (x* - x_i) / M_ii - the
// linear solve with the right-hand side of row i moved so that x_i = x* (what a sequential sweep does with a limited step)
__device__ __forceinline__ void repair_sweep(unsigned char* lds, const d2* mg, int nb, int lane, const unsigned char* OLD) {
  const int i = lane & 31, h = lane >> 5, l = lane & 7;
  const unsigned char* fixv = lds + OFF_FIXV;
  const double* mgs = reinterpret_cast<const double*>(mg);
  unsigned char* cur = lds + OFF_CUR;
  double* G = reinterpret_cast<double*>(lds + OFF_G);
  const int* ptr = reinterpret_cast<const int*>(lds + OFF_PTR);
  for (int b = 0; b < nb; ++b) {
    const u4 meta = *reinterpret_cast<const u4*>(lds + OFF_SROW + (b * BS + i) * SROW + h * 16);
    d2 M[8];
    load_m(M, mg, b, i, h);
    const d2 v0 = *reinterpret_cast<const d2*>(cur + (meta.x & 0xFFFF)), v1 = *reinterpret_cast<const d2*>(cur + (meta.x >> 16));
    const d2 v2 = *reinterpret_cast<const d2*>(cur + (meta.y & 0xFFFF)), v3 = *reinterpret_cast<const d2*>(cur + (meta.y >> 16));
    const d2 v4 = *reinterpret_cast<const d2*>(cur + (meta.z & 0xFFFF)), v5 = *reinterpret_cast<const d2*>(cur + (meta.z >> 16));
    const d2 v6 = *reinterpret_cast<const d2*>(cur + (meta.w & 0xFFFF));
    const double gx = halves_sum(((v0.x + v1.x) + (v2.x + v3.x)) + ((v4.x + v5.x) + v6.x));
    const double gy = halves_sum(((v0.y + v1.y) + (v2.y + v3.y)) + ((v4.y + v5.y) + v6.y));
    if (h == 0) {
      G[i] = gx;
      G[BS + i] = gy;
    }
    asm volatile("" ::: "memory");
    const d2* gqx = reinterpret_cast<const d2*>(G + 16 * h);
    const d2* gqy = reinterpret_cast<const d2*>(G + BS + 16 * h);
    double ax0 = 0.0, ax1 = 0.0, ay0 = 0.0, ay1 = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const d2 ggx = gqx[t], ggy = gqy[t];
      ax0 = __builtin_fma(M[t].x, ggx.x, ax0);
      ax1 = __builtin_fma(M[t].y, ggx.y, ax1);
      ay0 = __builtin_fma(M[t].x, ggy.x, ay0);
      ay1 = __builtin_fma(M[t].y, ggy.y, ay1);
    }
    const double xx = halves_sum(ax0 + ax1), xy = halves_sum(ay0 + ay1);
    const int soff = meta.w >> 16;                        // (h = 1 lanes: the zero record)
    if (h == 0) *reinterpret_cast<d2*>(cur + soff) = d2{xx, xy};
    asm volatile("" ::: "memory");
    unsigned long long fm = __builtin_amdgcn_ballot_w64(h == 0 && fixv[soff >> 4] != 0);
    while (fm) {
      const int r = __ffsll((long long)fm) - 1;
      fm &= fm - 1;
      const int v = __builtin_amdgcn_readlane(soff, r) >> 4;
      const d2 xs = exact_vertex(lds, v, l, OLD);
      const d2 xt = *reinterpret_cast<const d2*>(cur + v * 16);
      const double kk = (double)(ptr[v + 1] - ptr[v]);   // 1 / M_rr
      const double ddx = (xs.x - xt.x) * kk, ddy = (xs.y - xt.y) * kk;
      const double mjr = mgs[(((size_t)(b * 2 + (r >> 4)) * 8 + ((r & 15) >> 1)) * 32 + i) * 2 + (r & 1)];   // M[i][r]
      if (h == 0 && i >= r) {
        d2 xc = *reinterpret_cast<const d2*>(cur + soff);
        xc.x = __builtin_fma(mjr, ddx, xc.x);
        xc.y = __builtin_fma(mjr, ddy, xc.y);
        if (i == r) xc = xs;
        *reinterpret_cast<d2*>(cur + soff) = xc;
      }
      asm volatile("" ::: "memory");
    }
  }
}

// a sweep in plain index order by one wave, every interior vertex on the exact update (in place: an update sees the new
// positions of everything before it) - the way out for a checked sweep that does not settle within MAXROUNDS repair rounds
__device__ __forceinline__ void sequential_sweep(unsigned char* lds, int n_int, int lane) {
  const uint16_t* ivert = reinterpret_cast<const uint16_t*>(lds + OFF_IVERT);
  unsigned char* cur = lds + OFF_CUR;
  for (int r = 0; r < n_int; ++r) {
    const int v = ivert[r];
    const d2 xs = exact_vertex(lds, v, lane & 7, cur);
    if (lane == 0) *reinterpret_cast<d2*>(cur + v * 16) = xs;
    asm volatile("" ::: "memory");
  }
}

#ifdef MDQ_LIN_TRACE
// debug build only: s_memtime deltas of thread 0 of mesh 0 at the phase boundaries of the set-up
__device__ long long mdq_lin_trace_buf[16];
#define LT_STAMP(k) { __syncthreads(); const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_lin_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
extern "C" int mdq_lin_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_lin_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_lin_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#else
#define LT_STAMP(k)
#endif

__global__ __launch_bounds__(LWG) void smooth_linear_kernel(int NV, int NT, double* coords, const int32_t* cells,
                                                            const int32_t* nv_, const int32_t* nt_, const int32_t* iters_,
                                                            const int32_t* rem, const int32_t* rstat, int iters_env,
                                                            double* mws, int64_t mstride, int32_t* redo, int32_t* stats) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // sweeps: given per environment, or - inside an env step - `iters_env` where a vertex was removed successfully
  // (Env2DAirfoil._check_mesh -> flow_solver.remesh -> smooth(50), flow_solver.py:236-237; nothing otherwise)
  const int S = iters_ ? iters_[b] : ((rem[b] >= 0 && rstat[b] == 0) ? iters_env : 0);
  if (S <= 0) {
    if (tid == 0) redo[b] = stats[3 * b] = stats[3 * b + 1] = stats[3 * b + 2] = 0;
    return;
  }
  const int nv = nv_[b], nt = nt_[b];
  double2* x = reinterpret_cast<double2*>(coords) + (int64_t)b * NV;
  const int32_t* tri = cells + (int64_t)b * NT * 3;
  int* ptr = reinterpret_cast<int*>(lds + OFF_PTR);
  uint32_t* inc = reinterpret_cast<uint32_t*>(lds + OFF_INC);
  uint16_t* rk = reinterpret_cast<uint16_t*>(lds + OFF_RK);
  uint16_t* ivert = reinterpret_cast<uint16_t*>(lds + OFF_IVERT);
  unsigned char* lmeta = lds + OFF_LMETA;
  uint16_t* kdeg = reinterpret_cast<uint16_t*>(lds + OFF_KDEG);
  int* misc = reinterpret_cast<int*>(lds + OFF_MISC);
  int* part = reinterpret_cast<int*>(lds + OFF_PART);
  uint32_t* tmp = reinterpret_cast<uint32_t*>(lds + OFF_TMP);
  uint16_t* nbl = reinterpret_cast<uint16_t*>(lds + OFF_NB);
  int* cnt = reinterpret_cast<int*>(lds + OFF_SROW + 3 * LNT * 4);      // (scratch behind the arrival-order lists)
  static_assert(3 * LNT * 4 + LNV * 4 <= LNV * SROW, "cnt scratch");
#ifdef MDQ_LIN_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  // ---- vertex -> cells
  for (int v = tid; v < LNV; v += LWG) cnt[v] = 0;
  if (tid < 3) misc[tid] = tid == 2 ? 1 : 0;
  if (tid < 32) reinterpret_cast<double*>(lds + OFF_R2K)[tid] = tid ? 1.0 / (2.0 * tid) : 0.0;
  __syncthreads();
  for (int t = tid; t < nt; t += LWG)
    for (int k = 0; k < 3; ++k) atomicAdd(&cnt[tri[3 * t + k]], 1);
  __syncthreads();
  scan_inclusive(cnt, part);
  for (int v = tid; v < LNV; v += LWG) ptr[v + 1] = cnt[v];
  if (tid == 0) ptr[0] = 0;
  __syncthreads();
  for (int v = tid; v < LNV; v += LWG) cnt[v] = 0;
  __syncthreads();
  for (int t = tid; t < nt; t += LWG) {
    const int vs[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k], a = vs[(k + 1) % 3], c = vs[(k + 2) % 3];
      const int q = ptr[v] + atomicAdd(&cnt[v], 1);
      tmp[q] = (uint32_t)a | ((uint32_t)c << 10) | ((uint32_t)t << 20);
    }
  }
  __syncthreads();
  LT_STAMP(0)
  // ---- per vertex: cells in ascending cell order (the fixed order of every later sum), distinct neighbours in order
  // of first appearance with their multiplicity, interior test (every neighbour seen exactly twice)
  for (int v = tid; v < LNV; v += LWG) {
    const int q0 = ptr[v], k = ptr[v + 1] - q0;
    bool interior = v < nv && k > 0 && k <= MAXNB;
    if (v < nv && k > MAXNB) misc[2] = 0;                    // (a vertex of more than 16 cells: careful walk)
    int nn = 0;
    if (interior) {
      for (int e = 0; e < k; ++e) {                          // rank sort by cell id
        const uint32_t my = tmp[q0 + e];
        int rank = 0;
        for (int j = 0; j < k; ++j) rank += (tmp[q0 + j] >> 20) < (my >> 20);
        inc[q0 + rank] = my;
      }
      for (int e = 0; e < k; ++e) {
        const uint32_t w = inc[q0 + e];
        for (int s = 0; s < 2; ++s) {
          const uint32_t id = s ? (w >> 10) & 0x3FF : w & 0x3FF;
          int f = -1;
          for (int j = 0; j < nn; ++j)
            if ((nbl[v * MAXNB + j] & 0x3FF) == id) f = j;
          if (f >= 0) {
            nbl[v * MAXNB + f] += 1 << 10;
          } else if (nn < MAXNB) {
            nbl[v * MAXNB + nn++] = (uint16_t)(id | (1 << 10));
          } else {
            interior = false;                                // more than 16 distinct neighbours: a boundary fan
          }
        }
      }
      for (int j = 0; j < nn; ++j) interior = interior && (nbl[v * MAXNB + j] >> 10) == 2;
      interior = interior && nn == k;
    }
    cnt[v] = interior ? 1 : 0;
  }
  __syncthreads();
  LT_STAMP(1)
  scan_inclusive(cnt, part);
  const int n_int = cnt[LNV - 1];
  const int nb = (n_int + BS - 1) / BS;
  __syncthreads();
  for (int v = tid; v < LNV; v += LWG) {
    const bool interior = cnt[v] != (v ? cnt[v - 1] : 0);
    rk[v] = interior ? (uint16_t)(cnt[v] - 1) : (uint16_t)0xFFFF;
    if (interior) ivert[cnt[v] - 1] = (uint16_t)v;
  }
  __syncthreads();
  for (int v = tid; v < nv; v += LWG)                          // cell entries of the fixed vertices: nothing to validate
    if (rk[v] == 0xFFFF)
      for (int q = ptr[v]; q < ptr[v + 1]; ++q) inc[q] = 0xFFFFFFFFu;
  // (tmp / cnt are dead from here on: the solver rows take their place)
  LT_STAMP(2)
  // ---- per interior rank: gather slots (every neighbour that is not a lower-numbered member of the own block),
  // in-block lower neighbours (the strictly lower triangle of the block), validation flags of the cell entries
  for (int r = tid; r < nb * BS; r += LWG) {
    uint16_t slots[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) slots[s] = (uint16_t)ZOFF;
    int ns = 0, nl = 0, k = 1;
    if (r < n_int) {
      const int v = ivert[r], q0 = ptr[v];
      k = ptr[v + 1] - q0;
      for (int j = 0; j < k; ++j) {
        const int w = nbl[v * MAXNB + j] & 0x3FF;
        const int rw = rk[w];
        if (rw != 0xFFFF && w < v && (rw >> 5) == (r >> 5)) {
          if (nl < MAXLOW) lmeta[r * MAXLOW + nl] = (unsigned char)(rw & 31);
          ++nl;
        } else {
          // slot j of the row: lane half (j & 1), position (j >> 1)
          if (ns < NSLOT) {
            const int pos = (ns & 1) * 8 + (ns >> 1);
#pragma unroll
            for (int s = 0; s < 16; ++s)
              if (s == pos) slots[s] = (uint16_t)(w * 16);
          }
          ++ns;
        }
      }
      if (ns > NSLOT || nl > MAXLOW) misc[2] = 0;
      slots[7] = (uint16_t)(v * 16);                          // where the row's result goes
      for (int q = q0; q < q0 + k; ++q) {                      // validation entries: a | c << 10 | v << 20 | new(a) << 30 | new(c) << 31
        const uint32_t w = inc[q];
        const int a = w & 0x3FF, c = (w >> 10) & 0x3FF;
        inc[q] = (w & 0xFFFFF) | ((uint32_t)v << 20) | ((uint32_t)(rk[a] != 0xFFFF && a < v) << 30) |
                 ((uint32_t)(rk[c] != 0xFFFF && c < v) << 31);
      }
    }
    kdeg[r] = (uint16_t)(k | (min(nl, MAXLOW) << 8));
    uint32_t wds[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) wds[s] = (uint32_t)slots[2 * s] | ((uint32_t)slots[2 * s + 1] << 16);
    *reinterpret_cast<u4*>(lds + OFF_SROW + r * SROW) = u4{wds[0], wds[1], wds[2], wds[3]};
    *reinterpret_cast<u4*>(lds + OFF_SROW + r * SROW + 16) = u4{wds[4], wds[5], wds[6], wds[7]};
  }
  __syncthreads();
  LT_STAMP(3)
  const bool eligible = misc[2] != 0 && n_int > 0;
  if (!eligible) {                               // the careful walk takes all the sweeps
    if (tid == 0) {
      redo[b] = S;
      stats[3 * b] = stats[3 * b + 1] = stats[3 * b + 2] = 0;
    }
    return;
  }
  // ---- block inverses: one wave per block, lane j = column j; row i of (I - N)^-1 = e_i + (1 / k_i) sum of the rows of
  // its in-block lower neighbours (rows are kept as packed lower triangles in LDS), then M = (I - N)^-1 D^-1 to the workspace
  double* mg = mws + (int64_t)b * mstride;
  {
    double* T = reinterpret_cast<double*>(lds + OFF_TRI) + wave * 528;
    const double* r2ktab = reinterpret_cast<const double*>(lds + OFF_R2K);   // 2 x 1 / (2 k) = 1 / k, correctly rounded
    for (int blk = wave < TRW ? wave : nb; blk < nb; blk += TRW) {
      const int r0 = blk * BS, j = lane & 31;
      // the 32 rows are a chain (row i needs the rows of its in-block lower neighbours), so what counts is the latency of ONE
      // row step: its metadata - degree, lower count and the up to 8 lower neighbours, the same for every lane - is loaded
      // once, one row per lane, and broadcast with v_readlane inside the loop (as LDS reads it was two dependent round
      // trips in front of the reads of T: 64 k of the set-up's 140 k cycles went into this phase)
      const uint32_t kdv = kdeg[r0 + j];
      const uint2 lmv = *reinterpret_cast<const uint2*>(lmeta + (r0 + j) * MAXLOW);
      if (lane < 32) {
        for (int i = 0; i < BS; ++i) {
          const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)kdv, i);
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)lmv.x, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)lmv.y, i);
          const int nl = kd >> 8;
          // (the reads of T stay in a loop of nl steps: fully unrolled to 8 unconditional reads the step was 2.5x slower -
          // a lone wave issues ~170 instructions per row then, and issue, not the LDS round trips, bounds the chain)
          double val = 0.0;
          for (int t = 0; t < nl; ++t) {
            const int w = (int)(((t < 4 ? lo : hi) >> (8 * (t & 3))) & 0xFFu);
            if (j <= w) val += T[w * (w + 1) / 2 + j];
          }
          val = val * (2.0 * r2ktab[kd & 0xFF]) + (i == j ? 1.0 : 0.0);
          if (j <= i) T[i * (i + 1) / 2 + j] = val;
          asm volatile("" ::: "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // write-out: lane = row i (both halves of the wave: columns 0..15 / 16..31), 16 bytes per store
      const int i = lane & 31, h = lane >> 5;
      d2* out = reinterpret_cast<d2*>(mg) + ((size_t)(blk * 2 + h) * 8) * 32 + i;
      for (int t = 0; t < 8; ++t) {
        const int j0 = 16 * h + 2 * t;
        const double m0 = j0 <= i ? T[i * (i + 1) / 2 + j0] * (2.0 * r2ktab[kdeg[r0 + j0] & 0xFF]) : 0.0;
        const double m1 = j0 + 1 <= i ? T[i * (i + 1) / 2 + j0 + 1] * (2.0 * r2ktab[kdeg[r0 + j0 + 1] & 0xFF]) : 0.0;
        out[t * 32] = d2{m0, m1};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __threadfence_block();
  __syncthreads();
  LT_STAMP(4)
  // ---- positions (the setup scratch is dead): cur and the snapshot of sweep 0; zero records
  for (int v = tid; v <= LNV; v += LWG) {
    d2 p = {0.0, 0.0};
    if (v < nv) {
      const double2 xv = x[v];
      p = d2{xv.x, xv.y};
    }
    *reinterpret_cast<d2*>(lds + OFF_CUR + v * 16) = p;
    *reinterpret_cast<d2*>(lds + OFF_SNAP + v * 16) = p;
    *reinterpret_cast<d2*>(lds + OFF_SNAP + PBUF + v * 16) = p;
  }
  __syncthreads();   // (also orders the workspace stores of this workgroup before its loads: same CU, same L2)
  LT_STAMP(5)
  // ---- sweeps.  Two modes:
  //  CHECKED (from sweep 0 until a sweep passes at the first try): solve, then ALL waves validate at once; vertices whose
  //    update was not clearly a full step are flagged and the sweep is redone from its snapshot by repair_sweep (exact
  //    update for the flagged vertices), until a round flags nothing new.  Right after a vertex removal the cavity's
  //    neighbours take limited steps in the first sweeps: one or two repair rounds in sweep 0, rarely later.
  //  PIPELINED: waves 0 / 1 solve sweep s WHILE waves 2, 3, 6, 7 validate sweep s - 1 from the two snapshots (waves 4 / 5
  //    stay off the solvers' SIMDs); a sweep that fails goes back to its snapshot and to the CHECKED mode.
  const d2* mgd = reinterpret_cast<const d2*>(mg);
  // validators of the pipelined mode: the waves of SIMDs 2 and 3 (waves 2, 3, 6, 7, 10, 11); the other waves of SIMDs 0 / 1
  // stay off the solvers' issue slots
  constexpr int NVAL = 2 * (LWG / 256);
  const int vw = (wave & 3) >= 2 ? 2 * (wave >> 2) + (wave & 1) : -1;
  const int ne = 3 * nt, npass = (ne + 63) / 64;
  unsigned char* fixv = lds + OFF_FIXV;
  auto copy_pos = [&](int dst_off, int src_off) {
    for (int v = tid; v < nv; v += LWG)
      *reinterpret_cast<d2*>(lds + dst_off + v * 16) = *reinterpret_cast<const d2*>(lds + src_off + v * 16);
  };
  auto solve_fast = [&]() {
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);
      solve_sweep<0>(lds, mgd, nb, lane);
      __builtin_amdgcn_s_setprio(0);
    } else if (wave == 1) {
      __builtin_amdgcn_s_setprio(3);
      solve_sweep<1>(lds, mgd, nb, lane);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  int s = 0, n_repaired = 0, n_rounds = 0, n_sentback = 0, n_seq = 0;   // (diagnostics)
  bool checked = true, pending = false;     // pending: sweep s - 1 still awaits its validation (pipelined mode)
  while (true) {
    if (s >= S && !pending) break;
    if (checked) {
      const int snap = OFF_SNAP + (s & 1) * PBUF;                 // state at the start of sweep s
      for (int v = tid; v < LNV + 16; v += LWG) fixv[v] = 0;
      int rounds = 0, nflag = 0;
      while (true) {
        __syncthreads();
        if (nflag == 0) {
          solve_fast();
        } else if (wave == 0) {
          repair_sweep(lds, mgd, nb, lane, lds + snap);
        }
        if (tid == 0) misc[1] = 0;
        __syncthreads();
        bool bad = false;
        for (int p = wave; p < npass; p += LWG / 64) bad |= validate_entry<true>(lds, 64 * p + lane, ne, lds + snap, lds + OFF_CUR);
        if (bad) misc[1] = 1;
        __syncthreads();
        if (!misc[1]) break;
        nflag = 1;
        ++n_rounds;
        copy_pos(OFF_CUR, snap);                                   // back to the start of the sweep
        if (++rounds >= MAXROUNDS) {                               // does not settle: this sweep in plain index order
          __syncthreads();
          if (wave == 0) sequential_sweep(lds, n_int, lane);
          __syncthreads();
          ++n_seq;
          break;
        }
      }
      copy_pos(OFF_SNAP + ((s + 1) & 1) * PBUF, OFF_CUR);
      __syncthreads();
      ++s;
      if (rounds == 0) {
        checked = false;
        pending = false;
      } else {
        ++n_repaired;
      }
    } else {
      if (s < S) solve_fast();
      bool bad = false;
      if (pending) {                            // sweep s - 1: OLD = snapshot (s - 1) & 1, NEW = snapshot s & 1
        const unsigned char* OLD = lds + OFF_SNAP + ((s - 1) & 1) * PBUF;
        const unsigned char* NEW = lds + OFF_SNAP + (s & 1) * PBUF;
        if (s < S) {
          if (vw >= 0)
            for (int p = vw; p < npass; p += NVAL) bad |= validate_entry<false>(lds, 64 * p + lane, ne, OLD, NEW);
        } else {                                // behind the last sweep: everybody
          for (int p = wave; p < npass; p += LWG / 64) bad |= validate_entry<false>(lds, 64 * p + lane, ne, OLD, NEW);
        }
      }
      if (tid == 0) misc[1] = 0;
      __syncthreads();
      if (bad) misc[1] = 1;
      __syncthreads();
      if (misc[1]) {                            // sweep s - 1 was not all full steps: redo it checked
        --s;
        ++n_sentback;
        copy_pos(OFF_CUR, OFF_SNAP + (s & 1) * PBUF);
        checked = true;
        pending = false;
        __syncthreads();
        continue;
      }
      if (s >= S) break;
      copy_pos(OFF_SNAP + ((s + 1) & 1) * PBUF, OFF_CUR);
      __syncthreads();
      pending = true;
      ++s;
    }
  }
  LT_STAMP(6)
  // ---- result
  __syncthreads();
  for (int v = tid; v < nv; v += LWG) {
    const d2 p = *reinterpret_cast<const d2*>(lds + OFF_CUR + v * 16);
    x[v] = double2{p.x, p.y};
  }
  if (tid == 0) {
    redo[b] = 0;
    stats[3 * b] = n_repaired;       // sweeps that took repair rounds (vertices with limited steps)
    stats[3 * b + 1] = n_rounds;     // repair rounds in total
    stats[3 * b + 2] = n_sentback | (n_seq << 16);   // pipelined sweeps whose validation failed (redone checked) | sweeps in plain order << 16
  }
}

// diagnostics: environments handed back to the careful walk by the last launches
__global__ void count_redo_kernel(int B, const int32_t* redo, unsigned long long* total) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && redo[b] > 0) atomicAdd(total, 1ull);
}
}  // namespace mdq_smooth_lin

extern "C" int64_t mdq_smooth_fast_workspace_bytes(int32_t B, int32_t NV) {
  if (B <= 0 || NV <= 0) return 0;
  if (NV > mdq_smooth_lin::LNV)       // a mesh beyond the 1024-vertex kernels: the tables of the level-scheduled kernel
    return smooth_big_workspace_bytes(B, NV, 1);
  const int64_t blocks = (NV + mdq_smooth_lin::BS - 1) / mdq_smooth_lin::BS + 2;     // two blocks of padding (prefetch)
  return (int64_t)B * blocks * mdq_smooth_lin::MBLK * 8 + (int64_t)B * 16 + 256;   // block inverses, [B] redo, [B][3] diagnostics
}

static int smooth_fast_impl(const char* who, int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells,
                            const int32_t* nv, const int32_t* nt, const int32_t* iterations, const int32_t* rem,
                            const int32_t* rstat, int32_t iters_env, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B <= 0 || !coords || !cells || !nv || !nt || (!iterations && (!rem || !rstat)) || !workspace)
    return mdq_set_error("mdq_smooth_fast: bad arguments");
  if (NV > mdq_smooth_lin::LNV || NT > mdq_smooth_lin::LNT)   // a mesh beyond the 1024-vertex kernels: level-scheduled kernel
    return smooth_big_launch(B, NV, NT, coords, cells, nv, nt, iterations, rem, rstat, iters_env, workspace, workspace_bytes, stream);
  if (workspace_bytes < mdq_smooth_fast_workspace_bytes(B, NV) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return mdq_set_error("mdq_smooth_fast: workspace too small or not 16-byte aligned (mdq_smooth_fast_workspace_bytes)");
  (void)who;
  const int64_t blocks = (NV + mdq_smooth_lin::BS - 1) / mdq_smooth_lin::BS + 2;
  const int64_t mstride = blocks * mdq_smooth_lin::MBLK;
  double* mws = reinterpret_cast<double*>(workspace);
  int32_t* redo = reinterpret_cast<int32_t*>(mws + (int64_t)B * mstride);
  hipStream_t st = (hipStream_t)stream;
  long long* trace = nullptr;
#ifdef MDQ_SMOOTH_TRACE
  trace = mdq_smooth_trace_host();
#endif
  // 1. every sweep as a blocked triangular solve: checked + repaired while limited steps occur, then validated in parallel
  hipLaunchKernelGGL(mdq_smooth_lin::smooth_linear_kernel, dim3(B), dim3(mdq_smooth_lin::LWG), 0, st, NV, NT, coords, cells, nv,
                     nt, iterations, rem, rstat, iters_env, mws, mstride, redo, redo + B);
  // 2. environments handed back (meshes beyond the kernel's limits: more than 16 cells at a vertex, 14 gather slots per
  //    row, 8 lower neighbours inside a block): the per-vertex walk, ONE workgroup over the - normally zero - environments
  //    with sweeps left (128 workgroups of 141 KB LDS each would wait for the other streams' kernels to leave the CUs)
  //    (four workgroups since round 4: a mesh family that hits the limits often would otherwise be walked one mesh after
  //    the other by a single workgroup; they return at once when nothing was handed back)
  hipLaunchKernelGGL(mdq_smoothing::smooth_kernel, dim3(B < 4 ? B : 4), dim3(mdq_smoothing::SWG), 0, st, B, NV, NT, coords, cells,
                     nv, nt, redo, 0, trace);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("mdq_smooth_fast: launch failed");
  return 0;
}

extern "C" int mdq_smooth_fast(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                               const int32_t* nt, const int32_t* iterations, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  if (!iterations) return mdq_set_error("mdq_smooth_fast: bad arguments");
  return smooth_fast_impl("mdq_smooth_fast", B, NV, NT, coords, cells, nv, nt, iterations, nullptr, nullptr, 0, workspace,
                          workspace_bytes, stream);
}

extern "C" int mdq_smooth_fast_env(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                                   const int32_t* nt, const int32_t* rem, const int32_t* rstat, int32_t iterations,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (!rem || !rstat) return mdq_set_error("mdq_smooth_fast_env: bad arguments");
  return smooth_fast_impl("mdq_smooth_fast_env", B, NV, NT, coords, cells, nv, nt, nullptr, rem, rstat, iterations, workspace,
                          workspace_bytes, stream);
}
