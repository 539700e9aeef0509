// Tile maps of the element-tile operator modes (5 / 7) built ON THE DEVICE from the dof <- element-slot lists.
//
// `IpcsBatch` (meshdqn_amd/ipcs_batch.py, topology.py `matfree_maps`) builds, on the host, for every chunk of 1 024
// consecutive triangles the ascending list of the rows the chunk touches (`mf_rlist`: row word | tile range) and, per
// triangle and local dof, the position of that dof in the chunk's row list and in the chunk's LDS tile (`mf_lpos`).  The
// S3 env step re-solves the flow on a mesh that exists only on the device (FlowSolver.remesh, flow_solver.py:233-359,
// after every Env2DAirfoil._remove_vertex): until round 6 its meshes beyond the LDS-resident modes ran WITHOUT tile maps -
// element results through 0.6 MB of global scratch per operator application and environment instead of the LDS tile.
// This kernel derives the same maps - bit for bit what the host builds for the same cells and dof numbering - from what
// mdq_env_topology already emits: g2_ptr / g2_src (dof <- slots e * 6 + i, ascending e).
//
// One 1 024-thread workgroup per environment.  A tile orders its entries by (row, triangle): the tile position of slot
// (e, i) with row r in chunk c is  T[c][r] + k,  T = exclusive prefix sum over the rows of the chunk's slot counts, k = the
// slot's index inside the row's run of chunk-c slots (the lists are ascending in e, so a chunk's slots of a row are one
// run).  Phases: (1) one thread per row walks its list and leaves the run length per chunk as a byte in LDS; (2) per chunk:
// exclusive scans of the counts (tile offsets) and of the touched flags (position in the row list) over the rows - serial
// inside a thread's contiguous range of rows, wave shuffles and one LDS exchange across threads -, the list entries are
// written, and the rows walk their run once more to write the packed words of their slots.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mdq_internal.h"
#include "meshdqn_hip.h"

namespace mdq_tm {

constexpr int WG = 1024, CH = 1024, NW = WG / 64;
constexpr int MAX_CH = 8;            // chunks of an environment (NT <= 8192)
constexpr int MAX_N2 = 16384;        // rows (NV <= 4096 vertices + NE <= 12288 edges)
constexpr int LMAX = 16;             // slots of a dof <- slot list (cells at a vertex)

__global__ __launch_bounds__(WG) void tile_maps_kernel(mdq_ipcs_desc d, int32_t* rlist, int32_t* rcnt, int32_t* lpos, int32_t* status) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = d.nv[b], nt = d.nt[b], n2 = nv + d.ne[b];
  const int nch = (nt + CH - 1) / CH, NCH = (d.NT + CH - 1) / CH, NRL = d.NRL;
  const int32_t* g2_ptr = d.g2_ptr + (int64_t)b * (d.N2 + 1);
  const int32_t* g2_src = d.g2_src + (int64_t)b * 6 * d.NT;
  int2* rl = reinterpret_cast<int2*>(rlist) + (int64_t)b * NCH * NRL;
  int32_t* rc = rcnt + (int64_t)b * NCH;
  int32_t* lp = lpos + (int64_t)b * 6 * d.NT;
  const int N2p = (d.N2 + 15) & ~15;
  unsigned char* cnt8 = smem;                                             // [nch][N2p] slots of row r in chunk c
  uint32_t* pk = reinterpret_cast<uint32_t*>(smem + (size_t)NCH * N2p);      // [N2p] position in the row list | tile offset << 16
  int* wtot = reinterpret_cast<int*>(pk + N2p);                           // [2][NW] wave totals of the two scans
  int* flag = wtot + 2 * NW;                                              // [1] failure of this environment
  if (tid == 0) flag[0] = 0;
  for (int k = tid; k < nch * N2p / 4; k += WG) reinterpret_cast<uint32_t*>(cnt8)[k] = 0u;
  __syncthreads();
  // ---- (1) run lengths per (chunk, row).  A list holds at most LMAX slots (a vertex of more than 16 cells is refused by the mesh
  //      kernels as well); all of them are requested at once: walked slot by slot, a row was a chain of dependent L2 round trips
  for (int r = tid; r < n2; r += WG) {
    const int lo = g2_ptr[r], hi = g2_ptr[r + 1];
    if (hi <= lo || hi - lo > LMAX) flag[0] = 1;  // a dof that belongs to no cell (the first-touch flags would miss its row) / too many
    int ent[LMAX];
#pragma unroll
    for (int k = 0; k < LMAX; ++k) ent[k] = lo + k < hi ? g2_src[lo + k] : -1;
    int c_prev = -1, run = 0;
#pragma unroll
    for (int k = 0; k < LMAX; ++k)
      if (ent[k] >= 0) {
        const int c = (ent[k] / 6) / CH;
        if (c != c_prev) {
          if (c_prev >= 0) cnt8[c_prev * N2p + r] = (unsigned char)run;
          if (c < c_prev || c >= nch) flag[0] = 1;   // (not ascending / beyond the mesh: not a list this kernel understands)
          c_prev = c;
          run = 0;
        }
        ++run;
      }
    if (c_prev >= 0) cnt8[c_prev * N2p + r] = (unsigned char)run;
  }
  __syncthreads();
  // ---- (2) per chunk: scans over the rows, list entries, packed words
  const int per = (n2 + WG - 1) / WG;              // contiguous rows per thread
  const int r0 = min(tid * per, n2), r1 = min(r0 + per, n2);
  for (int c = 0; c < nch; ++c) {
    const unsigned char* cc = cnt8 + c * N2p;
    int s_cnt = 0, s_tch = 0;
    for (int r = r0; r < r1; ++r) {
      const int k = cc[r];
      s_cnt += k;
      s_tch += k > 0;
    }
    int i_cnt = s_cnt, i_tch = s_tch;               // inclusive scans over the threads
    for (int off = 1; off < 64; off <<= 1) {
      const int a = __shfl_up(i_cnt, off), t = __shfl_up(i_tch, off);
      if (lane >= off) {
        i_cnt += a;
        i_tch += t;
      }
    }
    if (lane == 63) {
      wtot[wave] = i_cnt;
      wtot[NW + wave] = i_tch;
    }
    __syncthreads();
    int b_cnt = 0, b_tch = 0, t_tch = 0;
    for (int w = 0; w < NW; ++w) {
      if (w < wave) {
        b_cnt += wtot[w];
        b_tch += wtot[NW + w];
      }
      t_tch += wtot[NW + w];
    }
    int e_cnt = b_cnt + i_cnt - s_cnt, e_tch = b_tch + i_tch - s_tch;   // exclusive, at this thread's first row
    const bool fits = t_tch <= NRL && t_tch <= 0xFFFF;
    if (!fits && tid == 0) flag[0] = 1;             // more touched rows than the row lists (and the LDS stage) hold
    if (tid == 0) rc[c] = t_tch;
    if (fits)
      for (int r = r0; r < r1; ++r) {
        const int k = cc[r];
        if (k > 0) {
          bool first = true, last = true;
          for (int c2 = 0; c2 < c; ++c2) first = first && cnt8[c2 * N2p + r] == 0;
          for (int c2 = c + 1; c2 < nch; ++c2) last = last && cnt8[c2 * N2p + r] == 0;
          const uint32_t word = (uint32_t)r | (first ? 0x80000000u : 0u) | (last ? 0x40000000u : 0u);
          rl[(int64_t)c * NRL + e_tch] = make_int2((int)word, e_cnt | (k << 16));
          pk[r] = (uint32_t)e_tch | ((uint32_t)e_cnt << 16);
          e_cnt += k;
          e_tch += 1;
        }
      }
    __syncthreads();
    // the rows' slots of this chunk: position in the row list | (tile offset + index in the run) << 16
    if (fits)
      for (int r = tid; r < n2; r += WG) {
        const int k = cc[r];
        if (k > 0) {
          const int lo = g2_ptr[r], hi = g2_ptr[r + 1];
          const uint32_t w0 = pk[r];
          int ent[LMAX];
#pragma unroll
          for (int q = 0; q < LMAX; ++q) ent[q] = lo + q < hi ? g2_src[lo + q] : -1;
          int j = 0;
#pragma unroll
          for (int q = 0; q < LMAX; ++q) {
            const int s = ent[q], e = s / 6;
            if (s >= 0 && e / CH == c) {
              lp[(s - e * 6) * d.NT + e] = (int)((w0 & 0xFFFFu) | (((w0 >> 16) + (uint32_t)j) << 16));
              ++j;
            }
          }
        }
      }
    __syncthreads();
  }
  for (int c = nch + tid; c < NCH; c += WG) rc[c] = 0;
  __syncthreads();
  // an environment whose maps could not be built keeps the dof <- slot path: the kernels read rcnt[0] < 0 as "no tile maps"
  if (tid == 0) {
    if (flag[0]) rc[0] = -1;
    if (status) status[b] = flag[0] ? 1 : 0;
  }
}


// ------------------------------------------------------------------ spatial order of the cells of the flow engine's private meshes
//
// A tile application is cheap when the 1 024 triangles of a chunk share their rows: ~2 500 touched rows per chunk in a mesh
// generator's order, ~5 000 (of at most 6 144) in the order a red refinement + a few dozen cavity re-triangulations leave behind -
// more than the kernels' LDS stage holds, and every row then meets nearly every chunk.  The flow leg of the S3 step works on a
// PRIVATE copy of the meshes (handed over by the main topology kernel): this kernel sorts the cells of that copy along a Morton
// curve of their centroids (16 bits per axis over the mesh's bounding box, ties by cell id: unique 64-bit keys, so the order is
// a function of the mesh alone) and permutes `cells` and - when given - the main engine's cell dofs [6][NT] of the same cells
// alike.  Nothing but the summation order of the element loops depends on the order of the cells.
constexpr int SORT_MAX = 8192;

__device__ __forceinline__ uint32_t part1by1(uint32_t x) {
  x &= 0xFFFFu;
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

__global__ __launch_bounds__(WG) void sort_cells_kernel(int NV, int NT, const double* coords, const int32_t* nv_, const int32_t* nt_,
                                                        int32_t* cells, int32_t* cell_dofs) {
  __shared__ unsigned long long key[SORT_MAX];
  __shared__ double red[4 * NW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = nv_[b], nt = nt_[b];
  const double* X = coords + (int64_t)b * NV * 2;
  int32_t* C = cells + (int64_t)b * NT * 3;
  int32_t* D = cell_dofs ? cell_dofs + (int64_t)b * 6 * NT : nullptr;
  int N = 1024;
  while (N < nt) N <<= 1;
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int i = tid; i < nv; i += WG) {
    const double x = X[2 * i], y = X[2 * i + 1];
    lo[0] = fmin(lo[0], x);
    hi[0] = fmax(hi[0], x);
    lo[1] = fmin(lo[1], y);
    hi[1] = fmax(hi[1], y);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fmin(lo[c], __shfl_xor(lo[c], off));
      hi[c] = fmax(hi[c], __shfl_xor(hi[c], off));
    }
  if (lane == 0) {
    red[wave] = lo[0];
    red[NW + wave] = hi[0];
    red[2 * NW + wave] = lo[1];
    red[3 * NW + wave] = hi[1];
  }
  __syncthreads();
  for (int w = 0; w < NW; ++w) {
    lo[0] = fmin(lo[0], red[w]);
    hi[0] = fmax(hi[0], red[NW + w]);
    lo[1] = fmin(lo[1], red[2 * NW + w]);
    hi[1] = fmax(hi[1], red[3 * NW + w]);
  }
  const double sx = hi[0] > lo[0] ? 65536.0 / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? 65536.0 / (hi[1] - lo[1]) : 0.0;
  for (int e = tid; e < N; e += WG) {
    unsigned long long k = ~0ull;                       // (padding: behind every cell)
    if (e < nt) {
      const int a = C[3 * e], b_ = C[3 * e + 1], c = C[3 * e + 2];
      const double cx = (X[2 * a] + X[2 * b_] + X[2 * c]) * (1.0 / 3.0), cy = (X[2 * a + 1] + X[2 * b_ + 1] + X[2 * c + 1]) * (1.0 / 3.0);
      const uint32_t qx = (uint32_t)min(65535, max(0, (int)((cx - lo[0]) * sx))), qy = (uint32_t)min(65535, max(0, (int)((cy - lo[1]) * sy)));
      k = ((unsigned long long)(part1by1(qx) | (part1by1(qy) << 1)) << 16) | (unsigned long long)e;
    }
    key[e] = k;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < N / 2; t += WG) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p_ = i | j;     // the pair (i, i + j) of this pass
        const unsigned long long a = key[i], c = key[p_];
        const bool up = (i & k) == 0;
        if ((a > c) == up) {
          key[i] = c;
          key[p_] = a;
        }
      }
      __syncthreads();
    }
  // permute: everything is read before anything is written (a workgroup owns its environment)
  constexpr int PER = SORT_MAX / WG;
  int32_t cv[PER][3], dv[PER][6];
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int t = tid + m * WG;
    if (t < nt) {
      const int e = (int)(key[t] & 0xFFFFull);
#pragma unroll
      for (int i = 0; i < 3; ++i) cv[m][i] = C[3 * e + i];
      if (D) {
#pragma unroll
        for (int i = 0; i < 6; ++i) dv[m][i] = D[i * NT + e];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int t = tid + m * WG;
    if (t < nt) {
#pragma unroll
      for (int i = 0; i < 3; ++i) C[3 * t + i] = cv[m][i];
      if (D) {
#pragma unroll
        for (int i = 0; i < 6; ++i) D[i * NT + t] = dv[m][i];
      }
    }
  }
}

}  // namespace mdq_tm

extern "C" MDQ_API int mdq_ipcs_build_tile_maps(const mdq_ipcs_desc* d, int32_t* status, void* stream) {
  using namespace mdq_tm;
  if (!d || d->B <= 0) return mdq_set_error("mdq_ipcs_build_tile_maps: empty descriptor");
  if (!d->g2_ptr || !d->g2_src || !d->nv || !d->nt || !d->ne)
    return mdq_set_error("mdq_ipcs_build_tile_maps needs nv, nt, ne and the dof <- slot lists (g2_ptr, g2_src)");
  if (!d->mf_rlist || !d->mf_rcnt || !d->mf_lpos || d->NRL <= 0)
    return mdq_set_error("mdq_ipcs_build_tile_maps writes mf_rlist [B][NCH][NRL][2], mf_rcnt [B][NCH] and mf_lpos [B][6][NT]: all three and NRL > 0");
  const int NCH = (d->NT + CH - 1) / CH;
  if (NCH > MAX_CH || d->N2 > MAX_N2 || d->NRL > 0xFFFF)
    return mdq_set_error("mdq_ipcs_build_tile_maps: capacities beyond the kernel (NT <= 8192, N2 <= 16384, NRL <= 65535)");
  const int N2p = (d->N2 + 15) & ~15;
  const size_t lds = (size_t)NCH * N2p + sizeof(uint32_t) * N2p + sizeof(int) * (2 * NW + 4);
  if (lds > 160 * 1024) return mdq_set_error("mdq_ipcs_build_tile_maps: tables beyond the LDS");
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_maps_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return mdq_set_error(hipGetErrorString(attr));
  hipLaunchKernelGGL(tile_maps_kernel, dim3(d->B), dim3(WG), lds, (hipStream_t)stream, *d, const_cast<int32_t*>(d->mf_rlist),
                     const_cast<int32_t*>(d->mf_rcnt), const_cast<int32_t*>(d->mf_lpos), status);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : mdq_set_error(hipGetErrorString(e));
}

extern "C" MDQ_API int mdq_flow_sort_cells(int32_t B, int32_t NV, int32_t NT, const double* coords, const int32_t* nv, const int32_t* nt,
                                           int32_t* cells, int32_t* cell_dofs, void* stream) {
  using namespace mdq_tm;
  if (B <= 0 || !coords || !nv || !nt || !cells) return mdq_set_error("mdq_flow_sort_cells: coords, nv, nt and cells are needed");
  if (NT > SORT_MAX) return mdq_set_error("mdq_flow_sort_cells: NT beyond 8192");
  hipLaunchKernelGGL(sort_cells_kernel, dim3(B), dim3(WG), 0, (hipStream_t)stream, NV, NT, coords, nv, nt, cells, cell_dofs);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : mdq_set_error(hipGetErrorString(e));
}
