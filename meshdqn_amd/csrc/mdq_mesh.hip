// Mesh-side kernels of the environment step: point location + P2/P1 evaluation of the stored
// snapshots on a coarsened mesh (Env2DAirfoil._check_mesh, Env2DAirfoil.py:556-593:
// `v_func.interpolate(original_u)`, `p_func.interpolate(original_p)` and the vertex evaluations).
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_mesh {

// One thread per target point: walk the candidate cells of its grid bin (ascending cell id), take the
// first cell whose barycentric coordinates are all >= 0, otherwise the candidate with the smallest
// violation (extrapolation from the nearest cell, `allow_extrapolation=True`), then evaluate all
// snapshots with the P2 / P1 bases of that cell.
__global__ __launch_bounds__(256) void interpolate_kernel(mdq_interp_desc d) {
  const int b = blockIdx.y;
  const int npts = d.npts[b] + (d.npts_extra ? d.npts_extra[b] : 0);
  const int np1 = d.np1[b];
  const int64_t B = b;
  const double* pts = d.points + B * d.NP * 2;
  // sparse mode (see the header): blocks [0, nbA) walk the points - vertices every snapshot, edge midpoints the last one
  // (or none) - blocks [nbA, gridDim.x) the three edges of the airfoil-facet cells, the remaining snapshots
  const int nbA = d.sparse ? (d.NP + 255) / 256 : (int)gridDim.x;
  const bool passB = (int)blockIdx.x >= nbA;
  const int nB = passB ? 3 * d.naf[b] : 0;
  const int kfirst = passB ? ((int)blockIdx.x - nbA) * (int)blockDim.x + (int)threadIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int kstride = passB ? ((int)gridDim.x - nbA) * (int)blockDim.x : nbA * (int)blockDim.x;
  for (int kk = kfirst; kk < (passB ? nB : npts); kk += kstride) {
    int k = kk, s_lo = 0, s_hi = d.S;
    if (passB) {
      const int f = kk / 3, j = kk - 3 * f;
      const int cell = d.af_facets[(B * d.NAF + f) * 2];
      k = d.cell_dofs[(B * 6 + 3 + j) * d.NT + cell];
      if (d.sparse == 1) s_hi = d.S - 1;          // (the last snapshot of every edge midpoint is pass A's)
      if (k < np1 || k >= npts || s_hi <= s_lo) continue;
    } else if (d.sparse && k >= np1) {
      if (d.sparse == 2) continue;
      s_lo = d.S - 1;
    }
    const double px = pts[2 * k], py = pts[2 * k + 1];
    int gx = (int)floor((px - d.x0) * d.inv_hx), gy = (int)floor((py - d.y0) * d.inv_hy);
    gx = gx < 0 ? 0 : (gx >= d.gnx ? d.gnx - 1 : gx);
    gy = gy < 0 ? 0 : (gy >= d.gny ? d.gny - 1 : gy);
    const int bin = gy * d.gnx + gx;
    int best = -1;
    double bxi = 0.0, beta = 0.0, bviol = -1e300;
    if (d.src_cellrec) {
      // candidates in batches of CB: their ids in one round trip, their records (vertex 0 + Jinv, 48 bytes) in a second
      // one, then the same tests in the same order.  (Cell by cell - id, then its dofs, then the vertex - a point paid
      // three dependent L2 round trips per candidate: the whole kernel was the latency of its longest candidate list.)
      constexpr int CB = 6;
      const int s0 = d.bin_ptr[bin], s1 = d.bin_ptr[bin + 1];
      bool found = false;
      for (int sb = s0; sb < s1 && !found; sb += CB) {
        int cid[CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) cid[q] = d.bin_cells[min(sb + q, s1 - 1)];
        double2 r0[CB], r1[CB], r2[CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          const double2* rp = reinterpret_cast<const double2*>(d.src_cellrec + (int64_t)cid[q] * 6);
          r0[q] = rp[0];
          r1[q] = rp[1];
          r2[q] = rp[2];
        }
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          if (sb + q < s1 && !found) {
            const double dx = px - r0[q].x, dy = py - r0[q].y;
            const double xi = r1[q].x * dx + r1[q].y * dy, eta = r2[q].x * dx + r2[q].y * dy;
            const double l0 = 1.0 - xi - eta;
            double viol = fmin(fmin(l0, xi), eta);
            viol = viol < 0.0 ? viol : 0.0;
            if (viol > bviol) {  // strict: the first (lowest id) best candidate wins
              bviol = viol;
              best = cid[q];
              bxi = xi;
              beta = eta;
              if (viol == 0.0) found = true;
            }
          }
        }
      }
    } else
    for (int s = d.bin_ptr[bin]; s < d.bin_ptr[bin + 1]; ++s) {
      const int c = d.bin_cells[s];
      const int v0 = d.src_cell_dofs[0 * d.src_nt + c];
      const double dx = px - d.src_coords[2 * v0], dy = py - d.src_coords[2 * v0 + 1];
      const double j00 = d.src_geom[0 * d.src_nt + c], j01 = d.src_geom[1 * d.src_nt + c];
      const double j10 = d.src_geom[2 * d.src_nt + c], j11 = d.src_geom[3 * d.src_nt + c];
      // reference coordinates: [xi, eta] = J^-1 (x - x0); geom stores Jinv[c][a] (reference row, physical col)
      const double xi = j00 * dx + j01 * dy, eta = j10 * dx + j11 * dy;
      const double l0 = 1.0 - xi - eta;
      double viol = fmin(fmin(l0, xi), eta);
      viol = viol < 0.0 ? viol : 0.0;
      if (viol > bviol) {  // strict: the first (lowest id) best candidate wins
        bviol = viol;
        best = c;
        bxi = xi;
        beta = eta;
        if (viol == 0.0) break;
      }
    }
    const double l0 = 1.0 - bxi - beta, l1 = bxi, l2 = beta;
    double phi[6];
    phi[0] = l0 * (2.0 * l0 - 1.0);
    phi[1] = l1 * (2.0 * l1 - 1.0);
    phi[2] = l2 * (2.0 * l2 - 1.0);
    phi[3] = 4.0 * l1 * l2;
    phi[4] = 4.0 * l0 * l2;
    phi[5] = 4.0 * l0 * l1;
    int dof[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dof[i] = d.src_cell_dofs[i * d.src_nt + best];
    for (int s = s_lo; s < s_hi; ++s) {
      const double* us = d.src_u + (int64_t)s * d.src_n2 * 2;
      double ux = 0.0, uy = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        ux += us[2 * dof[i]] * phi[i];
        uy += us[2 * dof[i] + 1] * phi[i];
      }
      double* uo = d.out_u + ((B * d.S + s) * d.NP + k) * 2;
      uo[0] = ux;
      uo[1] = uy;
      if (k < np1) {
        const double* ps = d.src_p + (int64_t)s * d.src_nv;
        d.out_p[(B * d.S + s) * d.NP1 + k] = ps[dof[0]] * l0 + ps[dof[1]] * l1 + ps[dof[2]] * l2;
      }
    }
    if (d.out_cell) d.out_cell[B * d.NP + k] = best;
  }
}

}  // namespace mdq_mesh

extern "C" int mdq_interpolate_snapshots(const mdq_interp_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->S <= 0 || d->NP <= 0) return mdq_set_error("mdq_interpolate_snapshots: bad arguments");
  int bx = (d->NP + 255) / 256;
  if (d->sparse) {
    if (d->sparse < 0 || d->sparse > 2 || !d->af_facets || !d->naf || !d->cell_dofs || d->NT <= 0 || d->NAF <= 0)
      return mdq_set_error("mdq_interpolate_snapshots: sparse mode needs af_facets, naf, cell_dofs, NT, NAF");
    bx += (3 * d->NAF + 255) / 256;
  }
  hipLaunchKernelGGL(mdq_mesh::interpolate_kernel, dim3(bx, d->B), dim3(256), 0, (hipStream_t)stream, *d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// State features of B environments (Env2DAirfoil.get_state, Env2DAirfoil.py:282-290), one thread per output value:
//   x[b][n][0:2]        = coordinates()[n_closest[n]]
//   x[b][n][2:2+2S]     = velocities[:, n_closest, :].reshape(N, -1)   (raw row-major reshape of the (S,N,2) block)
//   x[b][n][2+2S:2+3S]  = pressures[:, n_closest][:, :, 0].T
// with the reference's quirk that n_closest (a rank inside the removable list) is used as a VERTEX index; rows
// n >= nsel[b] (fewer than N selectable vertices left) are zero.
namespace mdq_mesh {

__global__ __launch_bounds__(256) void state_features_kernel(int B, int N, int S, int NV, int NP, const double* coords,
                                                             const double* u, const double* p, const int32_t* n_closest,
                                                             const int32_t* nsel, float* x) {
  const int F = 2 + 3 * S;
  const int64_t total = (int64_t)B * N * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int n = (int)((i / F) % N);
    const int b = (int)(i / ((int64_t)F * N));
    const int32_t* nc = n_closest + (int64_t)b * N;
    float val = 0.f;
    if (n < nsel[b]) {
      if (f < 2) {
        val = (float)coords[((int64_t)b * NV + nc[n]) * 2 + f];
      } else if (f < 2 + 2 * S) {
        // element (n, f-2) of the reshaped block = flat index q of the (S, N, 2) array
        const int q = n * 2 * S + (f - 2);
        const int s = q / (2 * N), r = q - s * 2 * N, m = r >> 1, c = r & 1;
        val = (float)u[(((int64_t)b * S + s) * NP + nc[m]) * 2 + c];
      } else {
        const int s = f - 2 - 2 * S;
        val = (float)p[((int64_t)b * S + s) * NV + nc[n]];
      }
    }
    x[i] = val;
  }
}

}  // namespace mdq_mesh

namespace mdq_mesh {
constexpr int RESTORE_MAX = 16;
struct RestoreArgs {
  uint32_t* dst[RESTORE_MAX];
  const uint32_t* src[RESTORE_MAX];
  int64_t words[RESTORE_MAX];
};
// blockIdx.x = position in the index list, blockIdx.y = tensor, blockIdx.z = 16 KB chunk of the row:
// one cached row -> row idx of that tensor, 16 bytes per thread and step where the row allows it
constexpr int RESTORE_CHUNK = 4096;   // words per block
__global__ __launch_bounds__(256) void restore_rows_kernel(RestoreArgs a, const int32_t* idx) {
  const int t = blockIdx.y;
  const int64_t n = a.words[t];
  const int64_t w0 = (int64_t)blockIdx.z * RESTORE_CHUNK;
  if (w0 >= n) return;
  const int64_t w1 = w0 + RESTORE_CHUNK < n ? w0 + RESTORE_CHUNK : n;
  uint32_t* d = a.dst[t] + (int64_t)idx[blockIdx.x] * n;
  const uint32_t* s = a.src[t];
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
    uint4* d4 = reinterpret_cast<uint4*>(d);
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    for (int64_t i = (w0 >> 2) + threadIdx.x; i < (w1 >> 2); i += 256) d4[i] = s4[i];
  } else {
    for (int64_t i = w0 + threadIdx.x; i < w1; i += 256) d[i] = s[i];
  }
}
}  // namespace mdq_mesh

extern "C" int mdq_restore_rows(int32_t n, void* const* dst, const void* const* src, const int64_t* row_bytes,
                                int32_t n_idx, const int32_t* idx, void* stream) {
  if (n <= 0 || n > mdq_mesh::RESTORE_MAX || !dst || !src || !row_bytes || n_idx < 0 || (n_idx > 0 && !idx))
    return mdq_set_error("mdq_restore_rows: bad arguments (at most 16 tensors)");
  if (n_idx == 0) return 0;
  mdq_mesh::RestoreArgs a;
  for (int t = 0; t < n; ++t) {
    if (!dst[t] || !src[t] || row_bytes[t] <= 0 || (row_bytes[t] & 3) || ((uintptr_t)dst[t] & 3) || ((uintptr_t)src[t] & 3))
      return mdq_set_error("mdq_restore_rows: rows must be non-empty, 4-byte aligned multiples of 4 bytes");
    a.dst[t] = static_cast<uint32_t*>(dst[t]);
    a.src[t] = static_cast<const uint32_t*>(src[t]);
    a.words[t] = row_bytes[t] / 4;
  }
  int64_t wmax = 0;
  for (int t = 0; t < n; ++t) wmax = a.words[t] > wmax ? a.words[t] : wmax;
  const int chunks = (int)((wmax + mdq_mesh::RESTORE_CHUNK - 1) / mdq_mesh::RESTORE_CHUNK);
  hipLaunchKernelGGL(mdq_mesh::restore_rows_kernel, dim3(n_idx, n, chunks), dim3(256), 0, (hipStream_t)stream, a, idx);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("restore_rows_kernel launch failed");
  return 0;
}

namespace mdq_mesh {
// up to 8 (strided) buffer copies in ONE launch: blockIdx.y = buffer, blockIdx.x walks its rows x 16 KB chunks
struct CopyArgs {
  uint32_t* dst[8];
  const uint32_t* src[8];
  int64_t rows[8], row_words[8], src_stride[8], dst_stride[8];   // strides in words
};
__global__ __launch_bounds__(256) void copy_strided_kernel(CopyArgs a) {
  const int t = blockIdx.y;
  const int64_t rw = a.row_words[t];
  const int64_t chunks = (rw + RESTORE_CHUNK - 1) / RESTORE_CHUNK;
  for (int64_t job = blockIdx.x; job < a.rows[t] * chunks; job += gridDim.x) {
    const int64_t r = job / chunks, c = job - r * chunks;
    const int64_t w0 = c * RESTORE_CHUNK, w1 = w0 + RESTORE_CHUNK < rw ? w0 + RESTORE_CHUNK : rw;
    uint32_t* d = a.dst[t] + r * a.dst_stride[t];
    const uint32_t* s = a.src[t] + r * a.src_stride[t];
    if (((rw | a.dst_stride[t] | a.src_stride[t]) & 3) == 0 && ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
      uint4* d4 = reinterpret_cast<uint4*>(d);
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      for (int64_t i = (w0 >> 2) + threadIdx.x; i < (w1 >> 2); i += 256) d4[i] = s4[i];
    } else {
      for (int64_t i = w0 + threadIdx.x; i < w1; i += 256) d[i] = s[i];
    }
  }
}
}  // namespace mdq_mesh

extern "C" int mdq_copy_strided(int32_t n, void* const* dst, const void* const* src, const int64_t* rows,
                                const int64_t* row_bytes, const int64_t* src_stride_bytes, const int64_t* dst_stride_bytes,
                                void* stream) {
  if (n <= 0 || n > 8 || !dst || !src || !rows || !row_bytes || !src_stride_bytes || !dst_stride_bytes)
    return mdq_set_error("mdq_copy_strided: bad arguments (at most 8 buffers)");
  mdq_mesh::CopyArgs a;
  int64_t jobs = 1;
  for (int t = 0; t < 8; ++t) {
    const int q = t < n ? t : 0;
    if (!dst[q] || !src[q] || rows[q] <= 0 || row_bytes[q] <= 0 || ((row_bytes[q] | src_stride_bytes[q] | dst_stride_bytes[q]) & 3) ||
        (((uintptr_t)dst[q] | (uintptr_t)src[q]) & 3))
      return mdq_set_error("mdq_copy_strided: rows and strides must be non-empty, 4-byte aligned multiples of 4 bytes");
    a.dst[t] = static_cast<uint32_t*>(dst[q]);
    a.src[t] = static_cast<const uint32_t*>(src[q]);
    a.rows[t] = rows[q];
    a.row_words[t] = row_bytes[q] / 4;
    a.src_stride[t] = src_stride_bytes[q] / 4;
    a.dst_stride[t] = dst_stride_bytes[q] / 4;
    const int64_t j = a.rows[t] * ((a.row_words[t] + mdq_mesh::RESTORE_CHUNK - 1) / mdq_mesh::RESTORE_CHUNK);
    jobs = j > jobs ? j : jobs;
  }
  const int gx = (int)(jobs < 1024 ? jobs : 1024);
  hipLaunchKernelGGL(mdq_mesh::copy_strided_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("copy_strided_kernel launch failed");
  return 0;
}

namespace mdq_mesh {
// blockIdx.x = environment: its first edge_ptr[b+1] - edge_ptr[b] padded entries -> the packed lists
__global__ __launch_bounds__(256) void compact_edges_kernel(int EMAX, const int32_t* __restrict__ src_pad,
                                                             const int32_t* __restrict__ dst_pad,
                                                             const int32_t* __restrict__ edge_ptr, int32_t* __restrict__ esrc,
                                                             int32_t* __restrict__ edst) {
  const int b = blockIdx.x, e0 = edge_ptr[b], n = edge_ptr[b + 1] - e0;
  const int32_t* s = src_pad + (int64_t)b * EMAX;
  const int32_t* d = dst_pad + (int64_t)b * EMAX;
  for (int i = threadIdx.x; i < n; i += 256) {
    esrc[e0 + i] = s[i];
    edst[e0 + i] = d[i];
  }
}
}  // namespace mdq_mesh

extern "C" int mdq_compact_edges(int32_t B, int32_t EMAX, const int32_t* src_pad, const int32_t* dst_pad,
                                 const int32_t* edge_ptr, int32_t* esrc, int32_t* edst, void* stream) {
  if (B <= 0 || EMAX <= 0 || !src_pad || !dst_pad || !edge_ptr || !esrc || !edst)
    return mdq_set_error("mdq_compact_edges: bad arguments");
  hipLaunchKernelGGL(mdq_mesh::compact_edges_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, EMAX, src_pad, dst_pad,
                     edge_ptr, esrc, edst);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("compact_edges_kernel launch failed");
  return 0;
}

extern "C" int mdq_state_features(int32_t B, int32_t N, int32_t S, int32_t NV, int32_t NP, const double* coords,
                                  const double* u, const double* p, const int32_t* n_closest, const int32_t* nsel,
                                  float* x, void* stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !coords || !u || !p || !n_closest || !nsel || !x)
    return mdq_set_error("mdq_state_features: bad arguments");
  const int64_t total = (int64_t)B * N * (2 + 3 * S);
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(mdq_mesh::state_features_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, N, S, NV, NP,
                     coords, u, p, n_closest, nsel, x);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("state_features_kernel launch failed");
  return 0;
}

// ================================================================== control logic of the batched env step on the device
//
// `VecEnv2DAirfoil.step` (Env2DAirfoil.step, Env2DAirfoil.py:318-377; calculate_reward :380-428) decides four things
// per environment: which vertex an action removes, how often the smoothing runs, reward + terminal flag, and which
// environments restart.  With these kernels the decisions stay on the device, so that a rollout is one uninterrupted
// stream of launches: no read-back between the Q-network forward and the next vertex removal.
namespace mdq_mesh {

// one wave per environment: greedy action = first maximum of q[b][0..NA) (torch.argmax), epsilon-greedy choice, then
// Env2DAirfoil.step's action decoding: N = "do nothing" (shift the N-closest window), an action without a vertex
// behind it = code 2 ("RAN OUT OF VERTICES"), otherwise the vertex id coord_map[b][action]
__global__ __launch_bounds__(64) void env_act_kernel(int N, const float* q, const uint8_t* explore, const int32_t* rand_action,
                                                      const int32_t* nsel, const int32_t* coord_map, int32_t* offset,
                                                      int32_t* action, int32_t* rem, int32_t* code) {
  const int b = blockIdx.x, lane = threadIdx.x, NA = N + 1;
  int a;
  if (q) {
    float best = -__builtin_inff();
    int bi = 0x7FFFFFFF;
    for (int i = lane; i < NA; i += 64) {
      const float v = q[(int64_t)b * NA + i];
      if (v > best || (v == best && i < bi) || (bi == 0x7FFFFFFF && !(v < best))) {   // (NaN rows: first index)
        best = v;
        bi = i;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    a = (explore && explore[b]) ? rand_action[b] : bi;
  } else {
    a = action[b];
  }
  if (lane == 0) {
    const bool shift = a == N;
    const bool pick = a >= 0 && a < nsel[b] && !shift;
    if (shift) offset[b] += 1;
    action[b] = a;
    rem[b] = pick ? coord_map[(int64_t)b * N + min(max(a, 0), N - 1)] : -1;
    code[b] = (!shift && !pick) ? 2 : 0;
  }
}

__global__ void env_smooth_iters_kernel(int B, const int32_t* rem, const int32_t* rstat, int32_t iterations, int32_t* its) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) its[b] = (rem[b] >= 0 && rstat[b] == 0) ? iterations : 0;
}

// reward / terminal flag (Env2DAirfoil.calculate_reward + the bookkeeping of step()): one thread per environment.
// Codes as in Env2DAirfoil.py:342-364: 0 = removed (reward from calculate_reward), 1 = "already removed" (reward -1, NOT
// terminal), 2 = broken / out of vertices (reward -1, terminal).  The reference's _remove_vertex / _check_mesh only ever
// return 0 or 2 (:458,:491,:573,:587,:598,:602) and nothing here produces 1 either; the branch is kept for parity.
__global__ void env_result_kernel(int B, int N, int S, const double* new_drags, const double* gt_drag, const int32_t* nv,
                                  int32_t nv0, const int32_t* rstat, const int32_t* topo_status, const int32_t* nsel,
                                  int32_t* code, int32_t* steps, double threshold, double time_reward, double goal_vertices,
                                  int32_t timesteps, double negative_reward, int32_t auto_reset, double* reward,
                                  uint8_t* done, int32_t* err_flag, int32_t* nv_out) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int c = code[b];
  if (rstat[b] != 0 || nsel[b] < N) c = 2;
  if (topo_status && topo_status[b] != 0) {
    c = 2;
    atomicOr(err_flag, 1);
  }
  const double drag_factor = -2.0 * log(0.5) / threshold;
  double ss = 0.0;
  bool acc = false;
  for (int s = 0; s < S; ++s) {
    const double g = gt_drag[s], d = new_drags[(int64_t)b * S + s];
    const double e = fabs(g - d) / fabs(g);
    ss += e * e;
    acc = acc || fabs(fabs(g - d) / g) > threshold;
  }
  const double drag_reward = 2.0 * exp(-drag_factor * sqrt(ss)) - 1.0;
  const double tr = (double)(nv0 - nv[b]) * time_reward;
  const bool vert = (double)nv[b] < goal_vertices * (double)nv0;
  const bool ok = c == 0;
  double r = ok ? drag_reward + tr : negative_reward;
  bool dn = ok ? (acc || vert) : (c != 1);
  const int st = steps[b] + 1;
  dn = dn || st >= timesteps;
  reward[b] = r;
  if (nv_out) nv_out[b] = nv[b];      // (the vertex count of the step, before an in-place reset rewrites nv)
  done[b] = dn ? 1 : 0;
  code[b] = c;
  steps[b] = (dn && auto_reset) ? 0 : st;
}

__global__ __launch_bounds__(256) void restore_rows_masked_kernel(RestoreArgs a, const uint8_t* mask) {
  if (!mask[blockIdx.x]) return;
  const int t = blockIdx.y;
  const int64_t n = a.words[t];
  const int64_t w0 = (int64_t)blockIdx.z * RESTORE_CHUNK;
  if (w0 >= n) return;
  const int64_t w1 = w0 + RESTORE_CHUNK < n ? w0 + RESTORE_CHUNK : n;
  uint32_t* d = a.dst[t] + (int64_t)blockIdx.x * n;
  const uint32_t* s = a.src[t];
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
    uint4* d4 = reinterpret_cast<uint4*>(d);
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    for (int64_t i = (w0 >> 2) + threadIdx.x; i < (w1 >> 2); i += 256) d4[i] = s4[i];
  } else {
    for (int64_t i = w0 + threadIdx.x; i < w1; i += 256) d[i] = s[i];
  }
}

// edge_ptr = exclusive prefix sums of nedges (one block; B <= 1024 per launch is plenty: environments per GPU)
__global__ __launch_bounds__(1024) void edge_ptr_kernel(int B, const int32_t* nedges, int32_t* edge_ptr) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  int run = 0;
  for (int base = 0; base < B; base += 1024) {
    const int i = base + tid;
    part[tid] = i < B ? nedges[i] : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    if (i < B) edge_ptr[i + 1] = run + part[tid];
    run += part[1023];
    __syncthreads();
  }
  if (tid == 0) edge_ptr[0] = 0;
}

}  // namespace mdq_mesh

extern "C" int mdq_env_act(int32_t B, int32_t N, const float* q, const uint8_t* explore, const int32_t* rand_action,
                           const int32_t* nsel, const int32_t* coord_map, int32_t* offset, int32_t* action, int32_t* rem,
                           int32_t* code, void* stream) {
  if (B <= 0 || N <= 0 || !nsel || !coord_map || !offset || !action || !rem || !code || (explore && !rand_action))
    return mdq_set_error("mdq_env_act: bad arguments");
  hipLaunchKernelGGL(mdq_mesh::env_act_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, N, q, explore, rand_action, nsel,
                     coord_map, offset, action, rem, code);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("env_act_kernel launch failed");
  return 0;
}

extern "C" int mdq_env_smooth_iters(int32_t B, const int32_t* rem, const int32_t* rstat, int32_t iterations, int32_t* its,
                                    void* stream) {
  if (B <= 0 || !rem || !rstat || !its) return mdq_set_error("mdq_env_smooth_iters: bad arguments");
  hipLaunchKernelGGL(mdq_mesh::env_smooth_iters_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, rem, rstat,
                     iterations, its);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("env_smooth_iters_kernel launch failed");
  return 0;
}

extern "C" int mdq_env_result(int32_t B, int32_t N, int32_t S, const double* new_drags, const double* gt_drag, const int32_t* nv,
                              int32_t nv0, const int32_t* rstat, const int32_t* topo_status, const int32_t* nsel,
                              int32_t* code, int32_t* steps, double threshold, double time_reward, double goal_vertices,
                              int32_t timesteps, double negative_reward, int32_t auto_reset, double* reward, uint8_t* done,
                              int32_t* err_flag, int32_t* nv_out, void* stream) {
  if (B <= 0 || S <= 0 || !new_drags || !gt_drag || !nv || !rstat || !nsel || !code || !steps || !reward || !done || !err_flag)
    return mdq_set_error("mdq_env_result: bad arguments");
  hipLaunchKernelGGL(mdq_mesh::env_result_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, B, N, S, new_drags,
                     gt_drag, nv, nv0, rstat, topo_status, nsel, code, steps, threshold, time_reward, goal_vertices, timesteps,
                     negative_reward, auto_reset, reward, done, err_flag, nv_out);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("env_result_kernel launch failed");
  return 0;
}

extern "C" int mdq_restore_rows_masked(int32_t n, void* const* dst, const void* const* src, const int64_t* row_bytes,
                                       int32_t B, const uint8_t* mask, void* stream) {
  if (n <= 0 || n > mdq_mesh::RESTORE_MAX || !dst || !src || !row_bytes || B <= 0 || !mask)
    return mdq_set_error("mdq_restore_rows_masked: bad arguments (at most 16 tensors)");
  mdq_mesh::RestoreArgs a;
  for (int t = 0; t < n; ++t) {
    if (!dst[t] || !src[t] || row_bytes[t] <= 0 || (row_bytes[t] & 3) || ((uintptr_t)dst[t] & 3) || ((uintptr_t)src[t] & 3))
      return mdq_set_error("mdq_restore_rows_masked: rows must be non-empty, 4-byte aligned multiples of 4 bytes");
    a.dst[t] = static_cast<uint32_t*>(dst[t]);
    a.src[t] = static_cast<const uint32_t*>(src[t]);
    a.words[t] = row_bytes[t] / 4;
  }
  int64_t wmax = 0;
  for (int t = 0; t < n; ++t) wmax = a.words[t] > wmax ? a.words[t] : wmax;
  const int chunks = (int)((wmax + mdq_mesh::RESTORE_CHUNK - 1) / mdq_mesh::RESTORE_CHUNK);
  hipLaunchKernelGGL(mdq_mesh::restore_rows_masked_kernel, dim3(B, n, chunks), dim3(256), 0, (hipStream_t)stream, a, mask);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("restore_rows_masked_kernel launch failed");
  return 0;
}

namespace mdq_mesh {
// The end of a device-resident env step in one launch: blockIdx.x = environment, blockIdx.y = one of gridDim.y workgroups
// that share its copy work (result logic: computed by every workgroup from the *_in arrays, written by y = 0).
// The terminal decision reads nv[b] and nsel[b], which are ALSO row arrays this launch resets in place: a workgroup (b, y)
// that started after another one had restored them would decide "not terminal" and skip its share of the rows - a
// half-reset environment (round 4's kernel: harmless only while all B Y workgroups were co-resident).  So the rows whose
// destination is d.nv / d.nsel are kept out of the strided restore: every workgroup of a resetting environment arrives
// at a counter when it has finished READING (decision + hand-over copies), and the LAST one to arrive restores them and
// puts the counter back to zero - no workgroup waits for another.
__global__ __launch_bounds__(256) void env_finish_kernel(mdq_env_finish_desc d) {
  const int b = blockIdx.x, y = blockIdx.y, Y = gridDim.y, tid = threadIdx.x;
  __shared__ int s_reset;
  if (tid == 0) {
#pragma clang fp contract(off)
    // ---- 1. mdq_env_result for environment b (same operations in the same order)
    int c = d.code_in[b];
    if (d.rstat[b] != 0 || d.nsel[b] < d.N) c = 2;
    if (d.topo_status && d.topo_status[b] != 0) {
      c = 2;
      if (y == 0) atomicOr(d.err_flag, 1);
    }
    const double drag_factor = -2.0 * log(0.5) / d.threshold;
    double ss = 0.0;
    bool acc = false;
    for (int s = 0; s < d.S; ++s) {
      const double g = d.gt_drag[s], dr = d.new_drags[(int64_t)b * d.S + s];
      const double e = fabs(g - dr) / fabs(g);
      ss += e * e;
      acc = acc || fabs(fabs(g - dr) / g) > d.threshold;
    }
    const double drag_reward = 2.0 * exp(-drag_factor * sqrt(ss)) - 1.0;
    const int nvb = d.nv[b];
    const double tr = (double)(d.nv0 - nvb) * d.time_reward;
    const bool vert = (double)nvb < d.goal_vertices * (double)d.nv0;
    const bool ok = c == 0;
    const double r = ok ? drag_reward + tr : d.negative_reward;
    bool dn = ok ? (acc || vert) : (c != 1);
    const int st = d.steps_in[b] + 1;
    dn = dn || st >= d.timesteps;
    if (y == 0) {
      d.reward[b] = r;
      if (d.nv_out) d.nv_out[b] = nvb;
      d.done[b] = dn ? 1 : 0;
      d.code_out[b] = c;
      d.steps_out[b] = (dn && d.auto_reset) ? 0 : st;
    }
    s_reset = (dn && d.auto_reset) ? 1 : 0;
  }
  __syncthreads();
  const bool reset = s_reset != 0;
  // ---- 2. hand-over of the pre-reset rows, then the in-place reset: word i of a row is read once, copied out if it lies
  // in the hand-over window, then overwritten (same thread: no ordering problem between the two)
  const int lin = y * 256 + tid, nlin = Y * 256;
  for (int t = 0; t < d.n_rows; ++t) {
    uint32_t* ho = static_cast<uint32_t*>(d.handover_dst[t]);
    const uint32_t* src = static_cast<const uint32_t*>(d.src[t]);
    const bool late = d.dst[t] == static_cast<const void*>(d.nv) || d.dst[t] == static_cast<const void*>(d.nsel);
    const bool restore = reset && src != nullptr && !late;
    if (!ho && !restore) continue;
    const int64_t words = d.row_bytes[t] >> 2, h0 = d.handover_off[t] >> 2, hw = d.handover_bytes[t] >> 2;
    uint32_t* row = static_cast<uint32_t*>(d.dst[t]) + (int64_t)b * words;
    uint32_t* hob = ho ? ho + (int64_t)b * hw : nullptr;
    const int64_t lo = restore ? 0 : h0, hi = restore ? words : h0 + hw;
    const bool vec = ((words | h0 | hw) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(row) | reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(hob)) & 15) == 0;
    if (vec) {
      uint4* row4 = reinterpret_cast<uint4*>(row);
      const uint4* src4 = reinterpret_cast<const uint4*>(src);
      uint4* ho4 = reinterpret_cast<uint4*>(hob);
      const int64_t h04 = h0 >> 2, h14 = (h0 + hw) >> 2;
      for (int64_t i = (lo >> 2) + lin; i < (hi >> 2); i += nlin) {
        if (ho && i >= h04 && i < h14) ho4[i - h04] = row4[i];
        if (restore) row4[i] = src4[i];
      }
    } else {
      for (int64_t i = lo + lin; i < hi; i += nlin) {
        if (ho && i >= h0 && i < h0 + hw) hob[i - h0] = row[i];
        if (restore) row[i] = src[i];
      }
    }
  }
  // ---- 2b. the decision's own inputs (nv, nsel): restored by the last workgroup of the environment to get here
  if (reset) {
    __syncthreads();                                   // this workgroup's hand-over reads are done
    if (tid == 0) {
      bool last = true;
      if (Y > 1) {
        __threadfence();
        last = atomicAdd(d.arrive + b, 1) == Y - 1;
      }
      if (last) {
        for (int t = 0; t < d.n_rows; ++t) {
          const uint32_t* src = static_cast<const uint32_t*>(d.src[t]);
          if (!src || (d.dst[t] != static_cast<const void*>(d.nv) && d.dst[t] != static_cast<const void*>(d.nsel))) continue;
          const int64_t words = d.row_bytes[t] >> 2;
          uint32_t* row = static_cast<uint32_t*>(d.dst[t]) + (int64_t)b * words;
          for (int64_t i = 0; i < words; ++i) row[i] = src[i];
        }
        if (Y > 1) d.arrive[b] = 0;                     // the next launch finds the counter at zero again
      }
    }
  }
  // ---- 3. node features of the next state (state_features_kernel's formulas); a reset environment: the cached ones
  const int N = d.N, S = d.S, F = 2 + 3 * S;
  float* xb = d.x + (int64_t)b * N * F;
  if (reset && d.x_init) {
    for (int i = lin; i < N * F; i += nlin) xb[i] = d.x_init[i];
    return;
  }
  if (reset) {               // (no cached features: read the rows restored above - needs all of them: one workgroup only)
    __threadfence_block();
    __syncthreads();
  }
  const int32_t* nc = d.n_closest + (int64_t)b * N;
  const int nselb = d.nsel[b];
  for (int i = lin; i < N * F; i += nlin) {
    const int n = i / F, f = i - n * F;
    float val = 0.f;
    if (n < nselb) {
      if (f < 2) {
        val = (float)d.coords[((int64_t)b * d.NV + nc[n]) * 2 + f];
      } else if (f < 2 + 2 * S) {
        const int q = n * 2 * S + (f - 2);
        const int s = q / (2 * N), r = q - s * 2 * N, m = r >> 1, c = r & 1;
        val = (float)d.u[(((int64_t)b * S + s) * d.NP + nc[m]) * 2 + c];
      } else {
        const int s = f - 2 - 2 * S;
        val = (float)d.p[((int64_t)b * S + s) * d.NV + nc[n]];
      }
    }
    xb[i] = val;
  }
}
}  // namespace mdq_mesh

extern "C" int mdq_env_finish(const mdq_env_finish_desc* d, void* stream) {
  if (!d || d->B <= 0 || d->N <= 0 || d->S <= 0 || d->n_rows < 0 || d->n_rows > MDQ_FINISH_MAX_ROWS || !d->new_drags ||
      !d->gt_drag || !d->nv || !d->rstat || !d->nsel || !d->code_in || !d->code_out || !d->steps_in || !d->steps_out ||
      d->steps_in == d->steps_out || !d->reward || !d->done || !d->err_flag || !d->coords || !d->u || !d->p || !d->n_closest || !d->x)
    return mdq_set_error("mdq_env_finish: bad arguments");
  int64_t bytes = 0;
  for (int t = 0; t < d->n_rows; ++t) {
    if (!d->dst[t] || d->row_bytes[t] <= 0 || ((d->row_bytes[t] | d->handover_off[t] | d->handover_bytes[t]) & 3) ||
        (((uintptr_t)d->dst[t] | (uintptr_t)d->src[t] | (uintptr_t)d->handover_dst[t]) & 3) ||
        (d->handover_dst[t] && (d->handover_off[t] < 0 || d->handover_bytes[t] <= 0 ||
                                d->handover_off[t] + d->handover_bytes[t] > d->row_bytes[t])))
      return mdq_set_error("mdq_env_finish: rows must be non-empty 4-byte aligned multiples of 4 bytes with a hand-over window inside them");
    bytes += d->handover_dst[t] ? d->handover_bytes[t] : 0;
  }
  // workgroups per environment: enough lanes for the hand-over copies AND the in-place reset of a terminated environment
  // (16 bytes per lane and pass, ~4 passes; the workgroups of the other environments find nothing to restore and leave);
  // without cached initial features the reset path needs the whole environment in one workgroup
  int64_t rbytes = 0;
  for (int t = 0; t < d->n_rows; ++t) rbytes += (d->auto_reset && d->src[t]) ? d->row_bytes[t] : 0;
  if (rbytes > bytes) bytes = rbytes;
  int Y = (int)((bytes / 16 + 1023) / 1024);
  Y = Y < 1 ? 1 : (Y > 16 ? 16 : Y);
  if (d->auto_reset && !d->x_init) Y = 1;
  if (d->auto_reset && Y > 1 && !d->arrive) {
    // (a caller without the counter array keeps one workgroup per environment: slower for large rows, never wrong)
    Y = 1;
  }
  hipLaunchKernelGGL(mdq_mesh::env_finish_kernel, dim3(d->B, Y), dim3(256), 0, (hipStream_t)stream, *d);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("env_finish_kernel launch failed");
  return 0;
}

extern "C" int mdq_edge_ptr(int32_t B, const int32_t* nedges, int32_t* edge_ptr, void* stream) {
  if (B <= 0 || !nedges || !edge_ptr) return mdq_set_error("mdq_edge_ptr: bad arguments");
  hipLaunchKernelGGL(mdq_mesh::edge_ptr_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, nedges, edge_ptr);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("edge_ptr_kernel launch failed");
  return 0;
}
