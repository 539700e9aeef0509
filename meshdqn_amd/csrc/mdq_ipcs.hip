// IPCS Navier-Stokes hot path for gfx950: assembly, time stepping, probes.
//
// One 512-thread workgroup (8 wave64) owns one environment for the whole
// launch: its operators, vectors and element scratch stay in that CU's L2 slice /
// LDS, every synchronisation is a workgroup barrier, every reduction a fixed
// tree (bitwise reproducible), and `nsteps` time steps run inside ONE launch.
//
// Reference semantics restated here (BaratiLab/MeshDQN):
//   flow_solver.py:98-120   UFL forms F1 / a2,L2 / a3,L3
//   flow_solver.py:123-144  Dirichlet BCs + SystemAssembler (symmetric elimination)
//   flow_solver.py:362-396  evolve()
//   probes.py:23-50         drag / lift surface integrals
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"
#include "mdq_device.h"
#include "mdq_elem.h"

// This file is compiled as several translation units (meshdqn_amd/build.py: -DMDQ_IPCS_PART=k), each instantiating the
// kernels of some operator modes together with their launchers - as ONE unit it took a minute of every rebuild:
//   part 0  entry points, assembly, matrix-free set-up, probes, the three kernels of mode 3
//   part 1  evolve_kernel<0 / 1> (assembled SELL operators)
//   part 2  evolve_kernel<5> (element tiles, global vectors), evolve_team_kernel (mode 4)
//   part 3  the kernels of mode 2 (element tiles, LDS / register vectors: the reproducible mode of the long runs)
// Without the macro everything is one unit.  Kernels and device tables have internal linkage: every unit that uses the
// reference-element tables owns a copy of them (ensure_tables).
#ifndef MDQ_IPCS_PART
#define MDQ_IPCS_PART -1
#endif
#define MDQ_IN_PART(k) (MDQ_IPCS_PART == (k) || MDQ_IPCS_PART == -1)

#ifndef MDQ_CORR_MX0_INTERLEAVE
#define MDQ_CORR_MX0_INTERLEAVE false
#endif
#ifndef MDQ_PCG_REGM
#define MDQ_PCG_REGM 1
#endif
#ifndef MDQ_SETUP_WG
#define MDQ_SETUP_WG 768
#endif
namespace mdq {

static __constant__ RefTab c_tab;

constexpr int MF_CH = 1024;  // triangles per LDS tile: 2 per thread, interleaved for FP64 ILP (host maps use the same chunking)

// ------------------------------------------------------------------ per-environment views

// mode 7: bytes that say which rows a workgroup's chunks touch, and which entries of the chunks' row lists are the FIRST
// touch of their row by that workgroup (a chunk's list holds at most min(N2, 6 MF_CH) rows), in doubles
__host__ __device__ inline int64_t work_touch_row_cap(int NV, int NE) {
  const int64_t N2 = (int64_t)NV + NE;
  return N2 < 6 * 1024 ? N2 : 6 * 1024;
}
__host__ __device__ inline int64_t work_touch_doubles(int NV, int NT, int NE) {
  const int64_t N2 = (int64_t)NV + NE, nch = ((int64_t)NT + 1023) / 1024;
  return (2 * N2 + nch * work_touch_row_cap(NV, NE) + 7) / 8 + 2;
}
__host__ __device__ inline int64_t work_per_env(int NV, int NT, int NE) {
  const int64_t N2 = (int64_t)NV + NE;
  // escr 12*NT | 6 velocity vectors (double2[N2]) | p_new[NV] | 24 spare | assembled modes: 5 history vectors + counter |
  // mode 5: the accumulation vector of the tile application (double2[N2]) | mode 7: the second workgroup's accumulation vector
  // | mode 7: touched-row bytes of the two workgroups (2 N2) + first-touch bytes of every chunk's row list
  const int64_t n = 12 * (int64_t)NT + 12 * N2 + NV + 26 + 10 * N2 + 2 + 2 * N2 + 2 + 2 * N2 + 2 + work_touch_doubles(NV, NT, NE);
  return (n + 31) & ~(int64_t)31;  // keep every environment's slab 256-byte aligned
}
// offset (in doubles, even) of the tentative-velocity history of the assembled modes 0-2 inside an environment's slab
__host__ __device__ inline int64_t work_hist_offset(int NV, int NT, int NE) {
  const int64_t N2 = (int64_t)NV + NE;
  return (12 * (int64_t)NT + 12 * N2 + NV + 24 + 1) & ~(int64_t)1;
}

// offset (in doubles, even) of mode 5's accumulation vector (behind the history vectors and their counter); mode 7's second
// one follows it at + 2 N2 + 2
__host__ __device__ inline int64_t work_ytmp_offset(int NV, int NT, int NE) {
  const int64_t N2 = (int64_t)NV + NE;
  return (work_hist_offset(NV, NT, NE) + 10 * N2 + 2 + 1) & ~(int64_t)1;
}

// offset (in doubles) of mode 7's touch bytes: behind the second accumulation vector
__host__ __device__ inline int64_t work_touch_offset(int NV, int NT, int NE) {
  const int64_t N2 = (int64_t)NV + NE;
  return work_ytmp_offset(NV, NT, NE) + 4 * N2 + 4;
}

// mode 5 with the chunk's input rows staged in LDS behind the element tile: does the stage (NRL rows) fit?
__host__ __device__ inline bool mode5_stage_fits(int NRL) {
  return NRL > 0 && 64 * sizeof(double) + sizeof(double2) * 6 * MF_CH + sizeof(double2) * (size_t)NRL <= 160 * 1024;
}
// LDS of the tile modes 5 / 7 behind the reduction scratch: the element tile (+ the stage of a chunk's input rows behind it when
// the packed local maps are given and fit) - what the launcher allocates and what the kernels may re-use in the pressure phase
__host__ __device__ inline size_t tile_lds_bytes(const mdq_ipcs_desc& d) {
  return sizeof(double2) * 6 * MF_CH +
         ((d.mf_lpos && d.mf_rlist && d.mf_rcnt && mode5_stage_fits(d.NRL)) ? sizeof(double2) * (size_t)d.NRL : 0);
}

struct EnvView {
  int nv, nt, ne, n2, nnz2, nnz1, naf;
  int NT;  // capacity (SoA stride of cell_dofs / geom)
  const double* coords;
  const int32_t* cell_dofs;  // [6][NT]
  const int8_t* cell_outflow;
  const int32_t *rowptr2, *colidx2, *asm2_ptr, *asm2_src;
  const int32_t *rowptr1, *colidx1, *asm1_ptr, *asm1_src;
  const int32_t *sl2_off, *sl2_col, *sl1_off, *sl1_col;
  const int32_t *mf_scat, *mf_tptr;
  int mf_tstride;  // N2+1
  const int32_t *mf_rlist, *mf_rcnt;   // mode 5: touched rows per chunk (or null)
  const int32_t* mf_lpos;              // mode 5: [6][NT] position of the dof in its chunk's row list | tile position << 16 (or null)
  int NRL, rl_flags;
  bool tiles;                          // modes 5 / 7: tile maps present (else: element scratch + dof <- slot lists)
  const int32_t *g2_ptr, *g2_src, *g1_ptr, *g1_src;
  const uint8_t* bcu_flag;
  const double* bcu_gx;
  const uint8_t* bcp_flag;
  const int32_t* af_facets;
  int nbo;
  const int32_t *bo_rows, *bo_ptr, *bo_col, *bo_src;
  double* bo_val;
  double* geom;  // [5][NT]
  double* A1;    // SELL [entries][4]
  double* Ms;
  double* K1s;
  double2* lift1;
  double2* lift3;
  double2* idiag1;
  double* sdiagM;
  double* sdiagK;
  double2* u_n;
  double* p_n;
  double* work;
#ifdef MDQ_PROFILE
  long long* sprof;  // LDS [8]: tile_accumulate phase cycles (thread 0)
#endif
};

__device__ __forceinline__ EnvView env_view(const mdq_ipcs_desc& d, int b) {
  EnvView v;
  v.nv = d.nv[b];
  v.nt = d.nt[b];
  v.ne = d.ne[b];
  v.n2 = v.nv + v.ne;
  v.naf = d.naf[b];
  v.NT = d.NT;
  const int64_t B = b;
  v.coords = d.coords + B * d.NV * 2;
  v.cell_dofs = d.cell_dofs + B * 6 * d.NT;
  v.cell_outflow = d.cell_outflow + B * d.NT;
  v.rowptr2 = d.rowptr2 + B * (d.N2 + 1);
  v.colidx2 = d.colidx2 + B * d.NNZ2;
  v.asm2_ptr = d.asm2_ptr + B * (d.NNZ2 + 1);
  v.asm2_src = d.asm2_src + B * 36 * d.NT;
  v.rowptr1 = d.rowptr1 + B * (d.NV + 1);
  v.colidx1 = d.colidx1 + B * d.NNZ1;
  v.asm1_ptr = d.asm1_ptr + B * (d.NNZ1 + 1);
  v.asm1_src = d.asm1_src + B * 9 * d.NT;
  v.sl2_off = d.sl2_off + B * (d.N2 / 64 + 2);
  v.sl2_col = d.sl2_col + B * d.NSE2;
  v.sl1_off = d.sl1_off + B * (d.NV / 64 + 2);
  v.sl1_col = d.sl1_col + B * d.NSE1;
  v.mf_scat = d.mf_scat + B * 6 * d.NT;
  v.mf_tstride = d.N2 + 1;
  v.mf_tptr = d.mf_tptr ? d.mf_tptr + B * ((d.NT + MF_CH - 1) / MF_CH) * (d.N2 + 1) : nullptr;
  v.NRL = d.NRL;
  v.rl_flags = d.rl_flags;
  v.mf_rlist = (d.mf_rlist && d.mf_rcnt && d.NRL > 0) ? d.mf_rlist + B * ((d.NT + MF_CH - 1) / MF_CH) * d.NRL * 2 : nullptr;
  v.mf_rcnt = v.mf_rlist ? d.mf_rcnt + B * ((d.NT + MF_CH - 1) / MF_CH) : nullptr;
  // (maps built on the device, mdq_ipcs_build_tile_maps: a first count < 0 marks an environment whose maps could not be built -
  //  it keeps the dof <- slot path)
  if (v.mf_rcnt && !d.mf_tptr && v.mf_rcnt[0] < 0) v.mf_rlist = v.mf_rcnt = nullptr;
  v.mf_lpos = (v.mf_rlist && d.mf_lpos && mode5_stage_fits(d.NRL)) ? d.mf_lpos + B * 6 * d.NT : nullptr;
  // the element results of an operator application go through the LDS tile: host-built maps (mf_scat + mf_tptr; the row lists
  // optional) or the row lists + packed local maps alone (what the device builds)
  v.tiles = v.mf_tptr != nullptr || (v.mf_rlist != nullptr && v.mf_lpos != nullptr);
  v.g2_ptr = d.g2_ptr + B * (d.N2 + 1);
  v.g2_src = d.g2_src + B * 6 * d.NT;
  v.g1_ptr = d.g1_ptr + B * (d.NV + 1);
  v.g1_src = d.g1_src + B * 3 * d.NT;
  v.bcu_flag = d.bcu_flag + B * d.N2;
  v.bcu_gx = d.bcu_gx + B * d.N2;
  v.bcp_flag = d.bcp_flag + B * d.NV;
  v.af_facets = d.af_facets + B * d.NAF * 2;
  v.nbo = d.nbo ? d.nbo[b] : 0;
  v.bo_rows = d.bo_rows + B * d.NBO;
  v.bo_ptr = d.bo_ptr + B * (d.NBO + 1);
  v.bo_col = d.bo_col + B * d.NBE;
  v.bo_src = d.bo_src + B * d.NBE;
  v.bo_val = d.bo_val + B * d.NBE * 4;
  v.geom = d.geom + B * 5 * d.NT;
  v.A1 = d.A1 + B * d.NSE2 * 4;
  v.Ms = d.Ms + B * d.NSE2;
  v.K1s = d.K1s + B * d.NSE1;
  v.lift1 = reinterpret_cast<double2*>(d.lift1) + B * d.N2;
  v.lift3 = reinterpret_cast<double2*>(d.lift3) + B * d.N2;
  v.idiag1 = reinterpret_cast<double2*>(d.idiag1) + B * d.N2;
  v.sdiagM = d.sdiagM + B * d.N2;
  v.sdiagK = d.sdiagK + B * d.NV;
  v.u_n = reinterpret_cast<double2*>(d.u_n) + B * d.N2;
  v.p_n = d.p_n + B * d.NV;
  v.work = d.work + B * work_per_env(d.NV, d.NT, d.NE);
  v.nnz2 = v.nnz1 = 0;  // (filled by the assembly kernel: light descriptors carry no patterns)
  return v;
}

__device__ __forceinline__ Geo load_geo(const EnvView& v, int e) {
  Geo g;
  g.j00 = v.geom[0 * v.NT + e];
  g.j01 = v.geom[1 * v.NT + e];
  g.j10 = v.geom[2 * v.NT + e];
  g.j11 = v.geom[3 * v.NT + e];
  g.det = v.geom[4 * v.NT + e];
  return g;
}

__device__ __forceinline__ void load_cell_coords(const EnvView& v, int e, double (&X)[3][2]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int vid = v.cell_dofs[k * v.NT + e];
    X[k][0] = v.coords[2 * vid];
    X[k][1] = v.coords[2 * vid + 1];
  }
}

// single basis function / gradient at a reference point without register arrays
__device__ __forceinline__ double p2_phi_one(int i, double l0, double l1, double l2) {
  if (i < 3) {
    const double l = sel3(i, l0, l1, l2);
    return l * (2.0 * l - 1.0);
  }
  const int k = i - 3;
  return 4.0 * sel3(k, l1, l0, l0) * sel3(k, l2, l2, l1);
}

__device__ __forceinline__ void p2_dphi_one(int j, double l0, double l1, double l2, double& dx, double& dy) {
  // reference gradients of the barycentric coordinates
  if (j < 3) {
    const double f = 4.0 * sel3(j, l0, l1, l2) - 1.0;
    dx = f * sel3(j, -1.0, 1.0, 0.0);
    dy = f * sel3(j, -1.0, 0.0, 1.0);
    return;
  }
  const int k = j - 3;
  const double la = sel3(k, l1, l0, l0), lb = sel3(k, l2, l2, l1);
  const double dax = sel3(k, 1.0, -1.0, -1.0), day = sel3(k, 0.0, -1.0, -1.0);
  const double dbx = sel3(k, 0.0, 0.0, 1.0), dby = sel3(k, 1.0, 1.0, 0.0);
  dx = 4.0 * (la * dbx + lb * dax);
  dy = 4.0 * (la * dby + lb * day);
}

// ================================================================== assembly

// B^{cd}_e[i][j] = int_{outflow facet k of cell e} phi_i (d_c phi_j) n_d ds
// (`- dot(mu*nabla_grad(U)*n, v)*ds`, flow_solver.py:109), 2-point Gauss (degree 3 integrand).
__device__ inline void outflow_entry(const EnvView& v, int e, int k, int i, int j, const Geo& g, double (&Bcd)[2][2]) {
  double X[3][2];
  load_cell_coords(v, e, X);
  const Facet f = facet_geometry(X, k);
  Bcd[0][0] = Bcd[0][1] = Bcd[1][0] = Bcd[1][1] = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double s = c_tab.gx[q], w = c_tab.gw[q] * f.len;
    const double xi = f.ra[0] + s * (f.rb[0] - f.ra[0]);
    const double eta = f.ra[1] + s * (f.rb[1] - f.ra[1]);
    const double l0 = 1.0 - xi - eta;
    const double phi = p2_phi_one(i, l0, xi, eta);
    double dx, dy;
    p2_dphi_one(j, l0, xi, eta, dx, dy);
    const double gxp = g.j00 * dx + g.j10 * dy;  // physical d/dx phi_j
    const double gyp = g.j01 * dx + g.j11 * dy;
    Bcd[0][0] += w * phi * gxp * f.nx;
    Bcd[0][1] += w * phi * gxp * f.ny;
    Bcd[1][0] += w * phi * gyp * f.nx;
    Bcd[1][1] += w * phi * gyp * f.ny;
  }
}

#if MDQ_IN_PART(0)
static __global__ __launch_bounds__(WG) void assemble_kernel(mdq_ipcs_desc d) {
  const int b = blockIdx.x;
  EnvView v = env_view(d, b);
  v.nnz2 = v.rowptr2[v.n2];
  v.nnz1 = v.rowptr1[v.nv];
  const int tid = threadIdx.x;
  const double a = d.rho / d.dt, mu = d.mu;

  // ---- phase 0: affine geometry per triangle
  for (int e = tid; e < v.nt; e += WG) {
    double X[3][2];
    load_cell_coords(v, e, X);
    const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0];
    const double J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
    const double det = J00 * J11 - J01 * J10;
    v.geom[0 * v.NT + e] = J11 / det;
    v.geom[1 * v.NT + e] = -J01 / det;
    v.geom[2 * v.NT + e] = -J10 / det;
    v.geom[3 * v.NT + e] = J00 / det;
    v.geom[4 * v.NT + e] = fabs(det);
  }
  __syncthreads();

  // position of entry j of row r inside the SELL-64 arrays
  auto pos2 = [&](int r, int j) { return v.sl2_off[r >> 6] + j * 64 + (r & 63); };
  auto pos1 = [&](int r, int j) { return v.sl1_off[r >> 6] + j * 64 + (r & 63); };

  // ---- phase 1: full (pre-BC) operator values, one thread per non-zero
  for (int k = tid; k < v.nnz2; k += WG) {
    double m = 0.0, kxx = 0.0, kxy = 0.0, kyx = 0.0, kyy = 0.0;
    double bxx = 0.0, bxy = 0.0, byx = 0.0, byy = 0.0;
    const int s0 = v.asm2_ptr[k], s1 = v.asm2_ptr[k + 1];
    int row = 0;
    for (int s = s0; s < s1; ++s) {
      const int slot = v.asm2_src[s];
      const int e = slot / 36, ij = slot - e * 36, i = ij / 6, j = ij - i * 6;
      if (s == s0) row = v.cell_dofs[i * v.NT + e];
      const Geo g = load_geo(v, e);
      m += g.det * c_tab.Mhat[i][j];
      // K^{ab}_ij = det * sum_{cd} Jinv[c][a] Jinv[d][b] Ghat[c][d][i][j]
      const double g00 = c_tab.Ghat[0][0][i][j], g01 = c_tab.Ghat[0][1][i][j];
      const double g10 = c_tab.Ghat[1][0][i][j], g11 = c_tab.Ghat[1][1][i][j];
      // Jinv[c][a]: c = reference index (row), a = physical index (col)
      const double Ja[2][2] = {{g.j00, g.j01}, {g.j10, g.j11}};
      kxx += g.det * (Ja[0][0] * (Ja[0][0] * g00 + Ja[1][0] * g01) + Ja[1][0] * (Ja[0][0] * g10 + Ja[1][0] * g11));
      kxy += g.det * (Ja[0][0] * (Ja[0][1] * g00 + Ja[1][1] * g01) + Ja[1][0] * (Ja[0][1] * g10 + Ja[1][1] * g11));
      kyx += g.det * (Ja[0][1] * (Ja[0][0] * g00 + Ja[1][0] * g01) + Ja[1][1] * (Ja[0][0] * g10 + Ja[1][0] * g11));
      kyy += g.det * (Ja[0][1] * (Ja[0][1] * g00 + Ja[1][1] * g01) + Ja[1][1] * (Ja[0][1] * g10 + Ja[1][1] * g11));
      const int ko = v.cell_outflow[e];
      if (ko >= 0) {
        double Bcd[2][2];
        outflow_entry(v, e, ko, i, j, g, Bcd);
        bxx += Bcd[0][0];
        bxy += Bcd[0][1];
        byx += Bcd[1][0];
        byy += Bcd[1][1];
      }
    }
    const double L = kxx + kyy;
    // block(c,d) = a M delta_cd + mu/2 (delta_cd L + K^{dc}) - mu/2 B^{cd}
    double4 blk;
    blk.x = a * m + 0.5 * mu * (L + kxx) - 0.5 * mu * bxx;
    blk.y = 0.5 * mu * kyx - 0.5 * mu * bxy;
    blk.z = 0.5 * mu * kxy - 0.5 * mu * byx;
    blk.w = a * m + 0.5 * mu * (L + kyy) - 0.5 * mu * byy;
    const int ps = pos2(row, k - v.rowptr2[row]);
    reinterpret_cast<double4*>(v.A1)[ps] = blk;
    v.Ms[ps] = m;
  }
  for (int k = tid; k < v.nnz1; k += WG) {
    double kk = 0.0;
    const int s0 = v.asm1_ptr[k], s1 = v.asm1_ptr[k + 1];
    int row = 0;
    for (int s = s0; s < s1; ++s) {
      const int slot = v.asm1_src[s];
      const int e = slot / 9, ij = slot - e * 9, i = ij / 3, j = ij - i * 3;
      if (s == s0) row = v.cell_dofs[i * v.NT + e];
      const Geo g = load_geo(v, e);
      // grad lambda_i = Jinv^T dl_i, dl = (-1,-1),(1,0),(0,1)
      const double dix = sel3(i, -g.j00 - g.j10, g.j00, g.j10), diy = sel3(i, -g.j01 - g.j11, g.j01, g.j11);
      const double djx = sel3(j, -g.j00 - g.j10, g.j00, g.j10), djy = sel3(j, -g.j01 - g.j11, g.j01, g.j11);
      kk += 0.5 * g.det * (dix * djx + diy * djy);
    }
    v.K1s[pos1(row, k - v.rowptr1[row])] = kk;
  }
  // outflow row list of the matrix-free mode 3 (same facet integrals as above, kept separate)
  if (v.nbo > 0) {
    const int nbe = v.bo_ptr[v.nbo];
    for (int t = tid; t < nbe; t += WG) {
      const int slot = v.bo_src[t];
      const int e = slot / 36, ij = slot - e * 36, i = ij / 6, j = ij - i * 6;
      const Geo g = load_geo(v, e);
      double Bcd[2][2];
      outflow_entry(v, e, v.cell_outflow[e], i, j, g, Bcd);
      reinterpret_cast<double4*>(v.bo_val)[t] = make_double4(Bcd[0][0], Bcd[0][1], Bcd[1][0], Bcd[1][1]);
    }
  }
  __syncthreads();

  // ---- phase 2: Dirichlet lifting vectors and diagonals, one thread per row
  for (int r = tid; r < v.n2; r += WG) {
    double l1x = 0.0, l1y = 0.0, l3x = 0.0, dx = 1.0, dy = 1.0, dm = 1.0;
    const bool fr = v.bcu_flag[r] != 0;
    const int k0 = v.rowptr2[r], len = v.rowptr2[r + 1] - k0;
    for (int j = 0; j < len; ++j) {
      const int c = v.colidx2[k0 + j], ps = pos2(r, j);
      const double4 blk = reinterpret_cast<const double4*>(v.A1)[ps];
      if (v.bcu_flag[c]) {
        const double gx = v.bcu_gx[c];  // gy = 0 for every BC of the reference
        l1x += blk.x * gx;
        l1y += blk.z * gx;
        l3x += v.Ms[ps] * gx;
      }
      if (c == r && !fr) {
        dx = blk.x;
        dy = blk.w;
        dm = v.Ms[ps];
      }
    }
    v.lift1[r] = make_double2(l1x, l1y);
    v.lift3[r] = make_double2(l3x, 0.0);
    v.idiag1[r] = make_double2(1.0 / dx, 1.0 / dy);
    v.sdiagM[r] = sqrt(dm);
  }
  for (int r = tid; r < v.nv; r += WG) {
    double dk = 1.0;
    if (!v.bcp_flag[r]) {
      const int k0 = v.rowptr1[r], len = v.rowptr1[r + 1] - k0;
      for (int j = 0; j < len; ++j)
        if (v.colidx1[k0 + j] == r) dk = v.K1s[pos1(r, j)];
    }
    v.sdiagK[r] = sqrt(dk);
  }
  __syncthreads();

  // ---- phase 3: symmetric elimination + Jacobi scaling (+ zero the SELL padding)
  const int rows2 = ((v.n2 + 63) >> 6) << 6;
  for (int r = tid; r < rows2; r += WG) {
    const int width = (v.sl2_off[(r >> 6) + 1] - v.sl2_off[r >> 6]) >> 6;
    int len = 0;
    if (r < v.n2) {
      const bool fr = v.bcu_flag[r] != 0;
      const double2 id = v.idiag1[r];
      const double sr = v.sdiagM[r];
      const int k0 = v.rowptr2[r];
      len = v.rowptr2[r + 1] - k0;
      for (int j = 0; j < len; ++j) {
        const int c = v.colidx2[k0 + j], ps = pos2(r, j);
        const bool fc = v.bcu_flag[c] != 0;
        double4 blk = reinterpret_cast<const double4*>(v.A1)[ps];
        double m = v.Ms[ps];
        if (fr || fc) {
          const double one = (c == r) ? 1.0 : 0.0;
          blk = make_double4(one, 0.0, 0.0, one);
          m = one;
        } else {
          blk.x *= id.x;
          blk.y *= id.x;
          blk.z *= id.y;
          blk.w *= id.y;
          m = m / (sr * v.sdiagM[c]);
        }
        reinterpret_cast<double4*>(v.A1)[ps] = blk;
        v.Ms[ps] = m;
      }
    }
    for (int j = len; j < width; ++j) {
      const int ps = pos2(r, j);
      reinterpret_cast<double4*>(v.A1)[ps] = make_double4(0.0, 0.0, 0.0, 0.0);
      v.Ms[ps] = 0.0;
    }
  }
  const int rows1 = ((v.nv + 63) >> 6) << 6;
  for (int r = tid; r < rows1; r += WG) {
    const int width = (v.sl1_off[(r >> 6) + 1] - v.sl1_off[r >> 6]) >> 6;
    int len = 0;
    if (r < v.nv) {
      const bool fr = v.bcp_flag[r] != 0;
      const double sr = v.sdiagK[r];
      const int k0 = v.rowptr1[r];
      len = v.rowptr1[r + 1] - k0;
      for (int j = 0; j < len; ++j) {
        const int c = v.colidx1[k0 + j], ps = pos1(r, j);
        const bool fc = v.bcp_flag[c] != 0;
        double kk = v.K1s[ps];
        if (fr || fc)
          kk = (c == r) ? 1.0 : 0.0;
        else
          kk = kk / (sr * v.sdiagK[c]);
        v.K1s[ps] = kk;
      }
    }
    for (int j = len; j < width; ++j) v.K1s[pos1(r, j)] = 0.0;
  }
}
#endif

// ================================================================== matrix-free operator setup
//
// Everything mode 3 (with the CG pressure solver) needs from a NEW mesh, without any global sparsity pattern:
// geometry, outflow-row blocks, Jacobi diagonals and Dirichlet lifting vectors of A1 / M accumulated row-wise from
// the element matrices (dof <- element-slot gathers g2/g1), the scaled + BC-eliminated P1 Laplacian in SELL-64.
// Same numbers as assemble_kernel up to the summation order.  One workgroup per environment.
#ifdef MDQ_SETUP_TRACE
static __device__ long long mdq_st_trace_buf[16];
#define ST_STAMP(k) { __syncthreads(); const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_st_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
#if MDQ_IN_PART(0)
extern "C" MDQ_API int mdq_st_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_st_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_st_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
#else
#define ST_STAMP(k)
#endif

// ST: the per-cell / per-dof data the row loops chase (geometry, cell dofs, Dirichlet flags and values, the P1 scaling) is
// staged in LDS first.  The row loops are chains of dependent loads (slot -> geometry / dofs -> flags -> values, per
// incident cell of every row): from global memory each level is an L2 round trip and the kernel was 270 k cycles of
// latency (P2 rows 134 k, P1 values 71 k, P1 diagonal 33 k); from LDS a level costs a tenth of that.
// 52 NT + 9 N2 + 9 NV bytes: 119 KB for ys930; a mesh that does not fit runs the unstaged instance.
constexpr int SWG = MDQ_SETUP_WG;   // threads of the set-up kernel (its row loops are latency chains: more rows in flight)
// LS (with ST): the dof <- element-slot lists (g1 / g2 pointers and slots, 16-bit in LDS) are staged as well when they fit:
// the row loops then have no dependent global load left.
// GEO = false with ST (round 6, the meshes whose 40 B of geometry per cell do not fit: the red-refined lab meshes): only the
// INDEX data is staged - cell dofs (16-bit), Dirichlet flags, the P1 scaling - 12 NT + N2 + 9 NV bytes (118 KB for the refined
// ys930); geometry and Dirichlet values come from global memory.  Of a slot's chain slot -> cell dofs -> flags (-> values at
// boundary cells only) the two inner levels are LDS round trips then; the unstaged instance paid an L2 round trip for each.
template <bool ST, bool LS = false, bool GEO = ST>
__global__ __launch_bounds__(SWG) void setup_matfree_kernel(mdq_ipcs_desc d) {
  static_assert(ST || !GEO, "geometry is staged together with the index data only");
  static_assert(GEO || !LS, "the slot lists are staged in the full instance only");
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const double a = d.rho / d.dt, mu = d.mu;
  extern __shared__ __align__(16) unsigned char setup_lds[];
  double* sGeo = reinterpret_cast<double*>(setup_lds);             // [5][NT]   (GEO)
  double* sGx = sGeo + (GEO ? 5 * (size_t)d.NT : 0);                // [N2]      (GEO)
  double* sSd = sGx + (GEO ? d.N2 : 0);                             // [NV]
  uint16_t* sCd = reinterpret_cast<uint16_t*>(sSd + d.NV);          // [6][NT]
  uint8_t* sFl = reinterpret_cast<uint8_t*>(sCd + 6 * (size_t)d.NT);  // [N2]
  uint8_t* sPf = sFl + d.N2;                                        // [NV]
  uint16_t* sG2p = reinterpret_cast<uint16_t*>(sPf + d.NV + ((d.N2 + d.NV) & 1));   // [N2 + 1]  (LS)
  uint16_t* sG2s = sG2p + d.N2 + 1;                                 // [6 NT]
  uint16_t* sG1p = sG2s + 6 * (size_t)d.NT;                         // [NV + 1]
  uint16_t* sG1s = sG1p + d.NV + 1;                                 // [3 NT]
  auto g2p = [&](int i) -> int { return LS ? (int)sG2p[i] : v.g2_ptr[i]; };
  auto g2s = [&](int i) -> int { return LS ? (int)sG2s[i] : v.g2_src[i]; };
  auto g1p = [&](int i) -> int { return LS ? (int)sG1p[i] : v.g1_ptr[i]; };
  auto g1s = [&](int i) -> int { return LS ? (int)sG1s[i] : v.g1_src[i]; };
  auto geo = [&](int e) -> Geo {
    if (!GEO) return load_geo(v, e);
    Geo g;
    g.j00 = sGeo[0 * v.NT + e];
    g.j01 = sGeo[1 * v.NT + e];
    g.j10 = sGeo[2 * v.NT + e];
    g.j11 = sGeo[3 * v.NT + e];
    g.det = sGeo[4 * v.NT + e];
    return g;
  };
  auto cdof = [&](int j, int e) -> int { return ST ? (int)sCd[j * v.NT + e] : v.cell_dofs[j * v.NT + e]; };
  auto uflag = [&](int i) -> bool { return ST ? sFl[i] != 0 : v.bcu_flag[i] != 0; };
  auto ugx = [&](int i) -> double { return GEO ? sGx[i] : v.bcu_gx[i]; };
  auto pflag = [&](int i) -> bool { return ST ? sPf[i] != 0 : v.bcp_flag[i] != 0; };
  auto sdk = [&](int i) -> double { return ST ? sSd[i] : v.sdiagK[i]; };
  if (ST) {
    for (int j = 0; j < 6; ++j)
      for (int e = tid; e < v.nt; e += SWG) sCd[j * v.NT + e] = (uint16_t)v.cell_dofs[j * v.NT + e];
    for (int i = tid; i < v.n2; i += SWG) {
      sFl[i] = v.bcu_flag[i];
      if (GEO) sGx[i] = v.bcu_gx[i];
    }
    for (int i = tid; i < v.nv; i += SWG) sPf[i] = v.bcp_flag[i];
    if (LS) {
      for (int i = tid; i <= v.n2; i += SWG) sG2p[i] = (uint16_t)v.g2_ptr[i];
      for (int i = tid; i < 6 * v.nt; i += SWG) sG2s[i] = (uint16_t)v.g2_src[i];
      for (int i = tid; i <= v.nv; i += SWG) sG1p[i] = (uint16_t)v.g1_ptr[i];
      for (int i = tid; i < 3 * v.nt; i += SWG) sG1s[i] = (uint16_t)v.g1_src[i];
    }
  }
  // reference-element tables in LDS: the row loops index them with the (lane-dependent) local row of a slot; from
  // constant memory that is one more dependent round trip per local column inside the conditional column loop
  __shared__ double sMhat[6][6];
  __shared__ double sGhat[2][2][6][6];
  for (int q = tid; q < 36; q += SWG) sMhat[q / 6][q % 6] = c_tab.Mhat[q / 6][q % 6];
  for (int q = tid; q < 144; q += SWG) sGhat[q / 72][(q / 36) % 2][(q / 6) % 6][q % 6] = c_tab.Ghat[q / 72][(q / 36) % 2][(q / 6) % 6][q % 6];
#ifdef MDQ_SETUP_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  for (int e = tid; e < v.nt; e += SWG) {
    double X[3][2];
    load_cell_coords(v, e, X);
    const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0];
    const double J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
    const double det = J00 * J11 - J01 * J10;
    const double q0 = J11 / det, q1 = -J01 / det, q2 = -J10 / det, q3 = J00 / det, q4 = fabs(det);
    v.geom[0 * v.NT + e] = q0;
    v.geom[1 * v.NT + e] = q1;
    v.geom[2 * v.NT + e] = q2;
    v.geom[3 * v.NT + e] = q3;
    v.geom[4 * v.NT + e] = q4;
    if (GEO) {
      sGeo[0 * v.NT + e] = q0;
      sGeo[1 * v.NT + e] = q1;
      sGeo[2 * v.NT + e] = q2;
      sGeo[3 * v.NT + e] = q3;
      sGeo[4 * v.NT + e] = q4;
    }
  }
  __syncthreads();
  ST_STAMP(0)
  if (v.nbo > 0) {
    const int nbe = v.bo_ptr[v.nbo];
    for (int t = tid; t < nbe; t += SWG) {
      const int slot = v.bo_src[t];
      const int e = slot / 36, ij = slot - e * 36, i = ij / 6, j = ij - i * 6;
      const Geo g = load_geo(v, e);
      double Bcd[2][2];
      outflow_entry(v, e, v.cell_outflow[e], i, j, g, Bcd);
      reinterpret_cast<double4*>(v.bo_val)[t] = make_double4(Bcd[0][0], Bcd[0][1], Bcd[1][0], Bcd[1][1]);
    }
  }
  ST_STAMP(1)
  // ---- P2 rows: raw diagonals (kept in idiag1 / sdiagM until the outflow rows have been corrected) and lifts
  for (int r = tid; r < v.n2; r += SWG) {
    double l1x = 0.0, l1y = 0.0, l3x = 0.0, dx = 0.0, dy = 0.0, dm = 0.0;
    // incident cells two at a time: the four dependent load levels of a cell (slot -> geometry / dofs -> Dirichlet
    // flags -> values) are paid once per pair; the second cell of an odd tail is the first one again, not added
    const int s0 = g2p(r), s1 = g2p(r + 1);
    for (int s = s0; s < s1; s += 2) {
      const bool two = s + 1 < s1;
      int ee[2], ii[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int slot = g2s((u == 1 && two) ? s + 1 : s);
        ee[u] = slot / 6;
        ii[u] = slot - ee[u] * 6;
      }
      Geo gg[2];
      int cj[2][6];
      bool fj[2][6];
      double gj[2][6];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        gg[u] = geo(ee[u]);
#pragma unroll
        for (int j = 0; j < 6; ++j) cj[u][j] = cdof(j, ee[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 6; ++j) fj[u][j] = uflag(cj[u][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 6; ++j) gj[u][j] = ugx(cj[u][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        const Geo g = gg[u];
        const int i = ii[u];
        const double Ja[2][2] = {{g.j00, g.j01}, {g.j10, g.j11}};
        // The diagonal term with the tables indexed by the lane's own i: ONE block for the wave.  (As a column loop
        // unrolled over j with `if (j != i && !fc) continue` the wave ran all six blocks - its lanes sit at different i -
        // i.e. ~270 fp64 instructions per cell for one term per lane.)  The Dirichlet columns follow in ascending j as
        // before; only waves that hold cells at the boundary enter those blocks.
        {
          const double m = g.det * sMhat[i][i];
          const double g00 = sGhat[0][0][i][i], g01 = sGhat[0][1][i][i];
          const double g10 = sGhat[1][0][i][i], g11 = sGhat[1][1][i][i];
          const double kxx = g.det * (Ja[0][0] * (Ja[0][0] * g00 + Ja[1][0] * g01) + Ja[1][0] * (Ja[0][0] * g10 + Ja[1][0] * g11));
          const double kyy = g.det * (Ja[0][1] * (Ja[0][1] * g00 + Ja[1][1] * g01) + Ja[1][1] * (Ja[0][1] * g10 + Ja[1][1] * g11));
          const double L = kxx + kyy;
          dx += a * m + 0.5 * mu * (L + kxx);
          dy += a * m + 0.5 * mu * (L + kyy);
          dm += m;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (!fj[u][j]) continue;
          const double m = g.det * sMhat[i][j];
          const double g00 = sGhat[0][0][i][j], g01 = sGhat[0][1][i][j];
          const double g10 = sGhat[1][0][i][j], g11 = sGhat[1][1][i][j];
          const double kxx = g.det * (Ja[0][0] * (Ja[0][0] * g00 + Ja[1][0] * g01) + Ja[1][0] * (Ja[0][0] * g10 + Ja[1][0] * g11));
          const double kxy = g.det * (Ja[0][0] * (Ja[0][1] * g00 + Ja[1][1] * g01) + Ja[1][0] * (Ja[0][1] * g10 + Ja[1][1] * g11));
          const double kyy = g.det * (Ja[0][1] * (Ja[0][1] * g00 + Ja[1][1] * g01) + Ja[1][1] * (Ja[0][1] * g10 + Ja[1][1] * g11));
          const double L = kxx + kyy;
          const double bx = a * m + 0.5 * mu * (L + kxx), bz = 0.5 * mu * kxy;
          const double gx = gj[u][j];
          l1x += bx * gx;
          l1y += bz * gx;
          l3x += m * gx;
        }
      }
    }
    v.lift1[r] = make_double2(l1x, l1y);
    v.lift3[r] = make_double2(l3x, 0.0);
    v.idiag1[r] = make_double2(dx, dy);
    v.sdiagM[r] = dm;
  }
  __syncthreads();
  ST_STAMP(2)
  // outflow rows: - mu/2 B on the diagonal and in the lifts (one thread per row: no conflicts)
  for (int t = tid; t < v.nbo; t += SWG) {
    const int row = v.bo_rows[t];
    double2 dg = v.idiag1[row], l1 = v.lift1[row];
    for (int k = v.bo_ptr[t]; k < v.bo_ptr[t + 1]; ++k) {
      const double4 bv = reinterpret_cast<const double4*>(v.bo_val)[k];
      const int c = v.bo_col[k];
      if (c == row) {
        dg.x -= 0.5 * mu * bv.x;
        dg.y -= 0.5 * mu * bv.w;
      }
      if (uflag(c)) {
        const double gx = ugx(c);
        l1.x -= 0.5 * mu * bv.x * gx;
        l1.y -= 0.5 * mu * bv.z * gx;
      }
    }
    v.idiag1[row] = dg;
    v.lift1[row] = l1;
  }
  __syncthreads();
  for (int r = tid; r < v.n2; r += SWG) {
    const bool fr = uflag(r);
    const double2 dg = v.idiag1[r];
    v.idiag1[r] = fr ? make_double2(1.0, 1.0) : make_double2(1.0 / dg.x, 1.0 / dg.y);
    v.sdiagM[r] = fr ? 1.0 : sqrt(v.sdiagM[r]);
  }
  ST_STAMP(3)
  // ---- P1 Laplacian: entry (r, c) = sum over the cells of r that contain c of |T| grad(l_r).grad(l_c)
  auto k1_entry = [&](int r, int c) {
    double kk = 0.0;
    for (int s = g1p(r); s < g1p(r + 1); ++s) {
      const int slot = g1s(s);
      const int e = slot / 3, i = slot - e * 3;
      int j = -1;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (cdof(q, e) == c) j = q;
      if (j < 0) continue;
      const Geo g = geo(e);
      const double dix = sel3(i, -g.j00 - g.j10, g.j00, g.j10), diy = sel3(i, -g.j01 - g.j11, g.j01, g.j11);
      const double djx = sel3(j, -g.j00 - g.j10, g.j00, g.j10), djy = sel3(j, -g.j01 - g.j11, g.j01, g.j11);
      kk += 0.5 * g.det * (dix * djx + diy * djy);
    }
    return kk;
  };
  for (int r = tid; r < v.nv; r += SWG) {
    const double sd = pflag(r) ? 1.0 : sqrt(k1_entry(r, r));
    v.sdiagK[r] = sd;
    if (ST) sSd[r] = sd;
  }
  __syncthreads();
  ST_STAMP(4)
  const int rows1 = ((v.nv + 63) >> 6) << 6;
  constexpr int KW = 16;   // widest row handled by the one-pass path (vertex degree + 1; wider rows: generic path)
  for (int r = tid; r < rows1; r += SWG) {
    const int off = v.sl1_off[r >> 6];
    const int width = (v.sl1_off[(r >> 6) + 1] - off) >> 6;
    if (width <= KW) {
      // ONE pass over the cells of the row: every cell adds its three entries (r, c_q) to the row's column list held in
      // registers (the generic path walks all cells of the row again for every entry: 7x the dependent loads; it
      // was 47 % of this kernel).  Same contributions in the same (cell) order per entry: same bits.
      int cols[KW];
      double vals[KW];
      int prev = -1;
#pragma unroll
      for (int j = 0; j < KW; ++j) {
        int c = -1;
        if (j < width && r < v.nv) {
          c = v.sl1_col[off + j * 64 + (r & 63)];
          if (c > prev) prev = c; else c = -1;   // (columns ascend; the padding repeats the row index)
        }
        cols[j] = c;
        vals[j] = 0.0;
      }
      if (r < v.nv) {
        if (pflag(r)) {
#pragma unroll
          for (int j = 0; j < KW; ++j) vals[j] = cols[j] == r ? 1.0 : 0.0;
        } else {
          for (int s_ = g1p(r); s_ < g1p(r + 1); ++s_) {
            const int slot = g1s(s_);
            const int e = slot / 3, i = slot - e * 3;
            const Geo g = geo(e);
            int cq[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) cq[q] = cdof(q, e);
            const double dix = sel3(i, -g.j00 - g.j10, g.j00, g.j10), diy = sel3(i, -g.j01 - g.j11, g.j01, g.j11);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const double djx = sel3(q, -g.j00 - g.j10, g.j00, g.j10), djy = sel3(q, -g.j01 - g.j11, g.j01, g.j11);
              const double val = 0.5 * g.det * (dix * djx + diy * djy);
#pragma unroll
              for (int j = 0; j < KW; ++j) vals[j] += cols[j] == cq[q] ? val : 0.0;
            }
          }
          const double sr = sdk(r);
#pragma unroll
          for (int j = 0; j < KW; ++j) {
            if (cols[j] >= 0) {
              const int c = cols[j];
              vals[j] = pflag(c) ? (c == r ? 1.0 : 0.0) : vals[j] / (sr * sdk(c));
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < KW; ++j)
        if (j < width) v.K1s[off + j * 64 + (r & 63)] = cols[j] >= 0 ? vals[j] : 0.0;
      continue;
    }
    int prev = -1;
    for (int j = 0; j < width; ++j) {
      const int ps = off + j * 64 + (r & 63);
      double kk = 0.0;
      if (r < v.nv) {
        const int c = v.sl1_col[ps];
        if (c > prev) {  // real entry (columns ascend; the padding repeats the row index)
          prev = c;
          if (pflag(r) || pflag(c))
            kk = (c == r) ? 1.0 : 0.0;
          else
            kk = k1_entry(r, c) / (sdk(r) * sdk(c));
        }
      }
      v.K1s[ps] = kk;
    }
  }
  ST_STAMP(5)
}

// ================================================================== element right-hand sides

struct ElemIdx {
  int dof[6];
};

__device__ __forceinline__ ElemIdx load_dofs(const EnvView& v, int e) {
  ElemIdx E;
#pragma unroll
  for (int i = 0; i < 6; ++i) E.dof[i] = v.cell_dofs[i * v.NT + e];
  return E;
}

// element loops of the three right-hand sides: one thread per triangle, results to the
// per-environment scratch (slot = cell*6+i / cell*3+j) that the dof-gather passes read
__device__ inline void rhs1_elements(const EnvView& v, const mdq_ipcs_desc& d, const double2* __restrict__ u,
                                     const double* __restrict__ p, double2* __restrict__ escr) {
  const double a = d.rho / d.dt;
  for (int e = threadIdx.x; e < v.nt; e += WG) {
    const ElemIdx E = load_dofs(v, e);
    const Geo g = load_geo(v, e);
    double2 ue[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ue[i] = u[E.dof[i]];
    double pe[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pe[i] = p[E.dof[i]];
    double2 r[6];
    elem_rhs1_vol(g, a, d.mu, d.rho, ue, pe, r);
    const int ko = v.cell_outflow[e];
    if (ko >= 0) {
      double X[3][2];
      load_cell_coords(v, e, X);
      elem_outflow_add(g, X, ko, 0.5 * d.mu, ue, r);  // + mu/2 <nabla_grad(u_n) n, v>
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) escr[e * 6 + i] = r[i];
  }
}

template <int NTH = WG>
__device__ inline void rhs2_elements(const EnvView& v, const mdq_ipcs_desc& d, const double2* __restrict__ u,
                                     const double* __restrict__ p, double* __restrict__ escr) {
  const double idt = 1.0 / d.dt;
  for (int e = threadIdx.x; e < v.nt; e += NTH) {
    const ElemIdx E = load_dofs(v, e);
    const Geo g = load_geo(v, e);
    double2 ue[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ue[i] = u[E.dof[i]];
    double pe[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pe[i] = p[E.dof[i]];
    double r[3];
    elem_rhs2(g, idt, ue, pe, r);
#pragma unroll
    for (int j = 0; j < 3; ++j) escr[e * 3 + j] = r[j];
  }
}

// The same right-hand side accumulated straight into an LDS vector with fp64 LDS atomics (mode 3): no element
// scratch in global memory and no per-row gather lists (a dependent chain of three global round trips per row and
// one more per incident cell: 36 % of the direct pressure kernel).  Two triangles per thread, their loads issued
// together.  `acc` must be zeroed and published by the caller.
template <int NTH>
__device__ inline void rhs2_accumulate_lds(const EnvView& v, const mdq_ipcs_desc& d, const double2* __restrict__ u,
                                           const double* __restrict__ p, double* acc) {
  const double idt = 1.0 / d.dt;
  for (int e0 = threadIdx.x; e0 < v.nt; e0 += 2 * NTH) {
    const int e1 = e0 + NTH;
    const bool two = e1 < v.nt;
    const int ec = two ? e1 : e0;   // (clamped: the second triangle's loads are always issued)
    const ElemIdx Ea = load_dofs(v, e0), Eb = load_dofs(v, ec);
    const Geo ga = load_geo(v, e0), gb = load_geo(v, ec);
    double2 ua[6], ub[6];
    double pa[3], pb[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      ua[i] = u[Ea.dof[i]];
      ub[i] = u[Eb.dof[i]];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      pa[i] = p[Ea.dof[i]];
      pb[i] = p[Eb.dof[i]];
    }
    double ra[3], rb[3];
    elem_rhs2(ga, idt, ua, pa, ra);
    elem_rhs2(gb, idt, ub, pb, rb);
#pragma unroll
    for (int j = 0; j < 3; ++j) unsafeAtomicAdd(acc + Ea.dof[j], ra[j]);
    if (two) {
#pragma unroll
      for (int j = 0; j < 3; ++j) unsafeAtomicAdd(acc + Eb.dof[j], rb[j]);
    }
  }
}

__device__ inline void rhs3_elements(const EnvView& v, const mdq_ipcs_desc& d, const double2* __restrict__ u,
                                     const double* __restrict__ pnew, const double* __restrict__ pold,
                                     double2* __restrict__ escr) {
  for (int e = threadIdx.x; e < v.nt; e += WG) {
    const ElemIdx E = load_dofs(v, e);
    const Geo g = load_geo(v, e);
    double2 ue[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ue[i] = u[E.dof[i]];
    double dp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dp[i] = pnew[E.dof[i]] - pold[E.dof[i]];
    double2 r[6];
    elem_rhs3(g, d.dt, ue, dp, r);
#pragma unroll
    for (int i = 0; i < 6; ++i) escr[e * 6 + i] = r[i];
  }
}

// ================================================================== matrix-free operator application
//
// y = Op x with the operator applied triangle by triangle (LDS-staged element tiles):
//   x   : gather vector in LDS (double2[n2])
//   es  : LDS tile of element results, SoA  es[i*CH + t]  (t = thread = triangle of the chunk)
// Per chunk of CH = WG triangles: every thread applies its triangle's 12x12 operator, barrier,
// then every thread sums the tile entries of its OWN rows (rows tid, tid+WG, ...) in ascending
// triangle order (fixed order => bitwise reproducible, no atomics), barrier.
// Requires n2 <= MF_ROWS*WG.
constexpr int MF_ROWS = 7;                     // own rows per thread: n2 <= MF_ROWS*WG = 3584

// Per-thread metadata of the triangles it applies in one chunk (MF_EPT = MF_CH / WG of them):
// 6 packed words (dof | tile position << 12 | (outflow edge + 1) << 28) + affine geometry.
// It is PREFETCHED one chunk ahead (rolling over to chunk 0 of the next application) so that the
// L2 latency of these loads hides behind the barrier / row-gather phase.
constexpr int MF_EPT = MF_CH / WG;
struct TileMeta {
  int w[MF_EPT][6];
  Geo g[MF_EPT];
};

__device__ __forceinline__ void tile_prefetch(const EnvView& v, TileMeta& tm, int chunk) {
#pragma unroll
  for (int j = 0; j < MF_EPT; ++j) {
    const int e = chunk * MF_CH + threadIdx.x + j * WG;
    if (e < v.nt) {
#pragma unroll
      for (int i = 0; i < 6; ++i) tm.w[j][i] = v.mf_scat[i * v.NT + e];
      tm.g[j] = load_geo(v, e);
    }
  }
}

// acc[k] = sum of the element results that land on own row tid + k*WG (zero for rows >= n2).
// op(e, geo, dofs, outflow_edge, ye) fills the 6 double2 results of triangle e.
// On entry tm holds the metadata of chunk 0; on exit again (prefetched for the next application).
template <class ElemOp>
__device__ __forceinline__ void tile_accumulate(const EnvView& v, double2* es, TileMeta& tm, ElemOp op,
                                                double2 (&acc)[MF_ROWS]) {
  const int tid = threadIdx.x, n = v.n2;
#pragma unroll
  for (int k = 0; k < MF_ROWS; ++k) acc[k] = make_double2(0.0, 0.0);
  const int nch = (v.nt + MF_CH - 1) / MF_CH;
#ifdef MDQ_PROFILE
  long long tq = __builtin_amdgcn_s_memtime();
#define MDQ_TSTAMP(k) { long long tn = __builtin_amdgcn_s_memtime(); if (tid == 0) v.sprof[k] += tn - tq; tq = tn; }
#else
#define MDQ_TSTAMP(k)
#endif
  for (int chunk = 0; chunk < nch; ++chunk) {
    double2* eb = es;
    // tile ranges of the own rows for this chunk (latency hides behind the element work)
    const int32_t* tp = v.mf_tptr + chunk * v.mf_tstride;
    int lohi[MF_ROWS];  // lo | hi << 16 (tile positions < 6*MF_CH < 65536)
#pragma unroll
    for (int k = 0; k < MF_ROWS; ++k) {
      const int row = tid + k * WG;
      lohi[k] = row < n ? (tp[row] | (tp[row + 1] << 16)) : 0;
    }
#pragma unroll
    for (int j = 0; j < MF_EPT; ++j) {
      const int e = chunk * MF_CH + tid + j * WG;
      if (e < v.nt) {
        ElemIdx E;
#pragma unroll
        for (int i = 0; i < 6; ++i) E.dof[i] = tm.w[j][i] & 0xFFF;
        double2 ye[6];
        op(e, tm.g[j], E, ((tm.w[j][0] >> 28) & 3) - 1, ye);
#pragma unroll
        for (int i = 0; i < 6; ++i) eb[(tm.w[j][i] >> 12) & 0x1FFF] = ye[i];
      }
    }
    MDQ_TSTAMP(0)
    tile_prefetch(v, tm, chunk + 1 < nch ? chunk + 1 : 0);
    MDQ_TSTAMP(1)
    __syncthreads();
    MDQ_TSTAMP(2)
#pragma unroll
    for (int k = 0; k < MF_ROWS; ++k) {
      for (int j = lohi[k] & 0xFFFF; j < (lohi[k] >> 16); ++j) {
        const double2 c = eb[j];
        acc[k].x += c.x;
        acc[k].y += c.y;
      }
    }
    MDQ_TSTAMP(3)
    __syncthreads();
    MDQ_TSTAMP(4)
  }
}

// MODE 5: the same element tiles for meshes whose vectors do not fit the LDS (ys930 red-refined: 12.9 k velocity dofs,
// 207 KB per vector): the operator input is gathered from GLOBAL memory (L2), the element results of a chunk go through the
// LDS tile, and the row owners (rows tid, tid + WG, ... as in every vector pass) add their tile entries - in fixed order -
// to an accumulation vector in the environment's workspace slab; after the last chunk every row hands its sum to `epi`.
// Per application and triangle: 6 dof ids + 6 tile positions + 5 geometry doubles + the outflow tag (89 B) instead of the
// 36 B per SELL entry of the assembled operator (~2 KB per triangle), and no sparsity pattern: like mode 2 it needs only
// what mdq_ipcs_setup_matfree derives on the device.  Bitwise reproducible.  mf_scat: packed words (N2 <= 4096) or plain
// tile positions (larger meshes, MeshTopology.matfree_maps).
#ifdef MDQ_T5_TRACE
// debug build only: s_memtime cycles of the phases of a mode-5 operator application (thread 0 of environment 0)
static __device__ long long mdq_t5_trace_buf[8];
#define T5_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_t5_trace_buf[k] += tn_ - t5q_; t5q_ = tn_; }
#if MDQ_IN_PART(2)
extern "C" MDQ_API int mdq_t5_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_t5_trace_buf), sizeof(long long) * 8) != hipSuccess) return -1;
  if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_t5_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
#else
#define T5_STAMP(k)
#endif

// The chunk loop of a tile application: chunks c0, c0 + cs, ... of the environment's triangles, their row sums accumulated
// into `ytmp` (c0 = 0, cs = 1: the whole operator; mode 7 deals the chunks out to the workgroups of a team, one accumulation
// vector each).  `flags_ok`: the touched-row lists mark the first chunk of every row IN THE ORDER 0, 1, 2, ... - usable only
// when this call walks all chunks in that order; otherwise the vector is zero-filled first.
template <class ElemOp>
__device__ __forceinline__ void tile_chunks_global(const EnvView& v, bool packed, double2* es, double2* ytmp, const double2* gx,
                                                   ElemOp op, int c0, int cs, bool flags_ok,
                                                   const unsigned char* firstb = nullptr, int first_stride = 0) {
  const int tid = threadIdx.x, n = v.n2;
  const int nch = (v.nt + MF_CH - 1) / MF_CH;
#ifdef MDQ_T5_TRACE
  long long t5q_ = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) mdq_t5_trace_buf[7] += 1;
#endif
  // rl_flags: the touched-row lists mark the first / last chunk of every row - no zero fill and no read at the first touch
  // (9.29 -> 9.07 ms per step of 128 refined meshes).  Running the caller's epilogue at the LAST touch and dropping the pass
  // over the rows at the end (two more vector streams saved) was measured SLOWER, 10.52 ms: the epilogue's own loads then
  // sit one row at a time inside the row loop instead of four rows in flight
  const bool rlf = flags_ok && v.mf_rlist && v.rl_flags != 0;
  // (firstb, mode 7: first-touch bytes of THIS walk's chunks - one per entry of a chunk's row list, see team_touch_setup)
  const bool fb = firstb != nullptr && v.mf_rlist;
  if (!rlf && !fb)
    for (int row = tid; row < n; row += WG) ytmp[row] = make_double2(0.0, 0.0);   // (own rows: visible to the row phases behind the barriers)
  double2* xst = es + 6 * MF_CH;                        // (staged input rows of the chunk: behind the tile, mf_lpos only)
  for (int chunk = c0; chunk < nch; chunk += cs) {
    if (v.mf_lpos) {
      const int2* rl = reinterpret_cast<const int2*>(v.mf_rlist) + (size_t)chunk * v.NRL;
      const int nr = v.mf_rcnt[chunk];
      for (int k = tid; k < nr; k += WG) xst[k] = gx[rl[k].x & 0x3FFFFFFF];
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < MF_EPT; ++j) {
      const int e = chunk * MF_CH + tid + j * WG;
      if (e < v.nt) {
        int pos[6];
        double2 xe[6];
        if (v.mf_lpos) {
          int w[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) w[i] = v.mf_lpos[i * v.NT + e];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            pos[i] = w[i] >> 16;
            xe[i] = xst[w[i] & 0xFFFF];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int w = v.mf_scat[i * v.NT + e];
            pos[i] = packed ? (w >> 12) & 0x1FFF : w;
            xe[i] = gx[v.cell_dofs[i * v.NT + e]];
          }
        }
        const Geo g = load_geo(v, e);
        double2 ye[6];
        op(e, g, (int)v.cell_outflow[e], xe, ye);
#pragma unroll
        for (int i = 0; i < 6; ++i) es[pos[i]] = ye[i];
      }
    }
    T5_STAMP(0)
    __syncthreads();
    T5_STAMP(1)
    if (v.mf_rlist) {
      // row phase over the rows this chunk TOUCHES (ascending list: one thread per entry, a row belongs to one entry per
      // chunk): entry and running sum of a batch are requested together.  Walking every own row's tile range (the branch
      // below) visited 12 924 rows per chunk on the refined mesh for ~2 500 touched ones: 57 % of an application.
      const int2* rl = reinterpret_cast<const int2*>(v.mf_rlist) + (size_t)chunk * v.NRL;
      const int nr = v.mf_rcnt[chunk];
#ifndef MDQ_T5_RB
#define MDQ_T5_RB 8
#endif
      constexpr int RB = MDQ_T5_RB;
      for (int k0 = tid; k0 < nr; k0 += RB * WG) {
        int2 en[RB];
        double2 a[RB];
        unsigned fmask = 0;
#pragma unroll
        for (int k = 0; k < RB; ++k) en[k] = rl[min(k0 + k * WG, nr - 1)];
        if (fb) {
          const unsigned char* fbc = firstb + (size_t)chunk * first_stride;
#pragma unroll
          for (int k = 0; k < RB; ++k) fmask |= fbc[min(k0 + k * WG, nr - 1)] ? 1u << k : 0u;
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          a[k] = make_double2(0.0, 0.0);
          if (!((rlf && en[k].x < 0) || ((fmask >> k) & 1u))) a[k] = ytmp[en[k].x & 0x3FFFFFFF];
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          if (k0 + k * WG < nr) {
            const int lo = en[k].y & 0xFFFF, cnt = en[k].y >> 16, row = en[k].x & 0x3FFFFFFF;
            for (int j = lo; j < lo + cnt; ++j) {
              const double2 c = es[j];
              a[k].x += c.x;
              a[k].y += c.y;
            }
            ytmp[row] = a[k];
          }
        }
      }
    } else {
      // row phase in batches of RB own rows: the tile ranges and the running sums of a batch are requested together
      const int32_t* tp = v.mf_tptr + chunk * v.mf_tstride;
      constexpr int RB = 8;
      for (int row0 = tid; row0 < n; row0 += RB * WG) {
        int lo[RB], hi[RB];
        double2 a[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          const int rc = min(row0 + k * WG, n - 1);
          lo[k] = tp[rc];
          hi[k] = tp[rc + 1];
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) a[k] = ytmp[min(row0 + k * WG, n - 1)];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
          const int row = row0 + k * WG;
          if (row < n && hi[k] > lo[k]) {
            for (int j = lo[k]; j < hi[k]; ++j) {
              const double2 c = es[j];
              a[k].x += c.x;
              a[k].y += c.y;
            }
            ytmp[row] = a[k];
          }
        }
      }
    }
    T5_STAMP(2)
    __syncthreads();
    T5_STAMP(3)
  }
}

// `gx`: the operator's input vector (global); op(e, geometry, outflow edge, xe[6], ye[6]) applies one triangle's operator to
// its six gathered input values.  With mf_lpos (round 5) the rows a chunk touches are STAGED in LDS behind the tile first -
// one pass over the chunk's ascending row list, coalesced where the rows are consecutive - and the triangles gather from
// there through 16-bit local indices (the triangle's dof ids + tile positions are one packed word per dof: 24 B per
// triangle instead of 48).
template <class ElemOp, class Epi>
__device__ __forceinline__ void tile_apply_global(const EnvView& v, bool packed, double2* es, double2* ytmp, const double2* gx,
                                                  ElemOp op, Epi epi) {
  const int tid = threadIdx.x, n = v.n2;
  if (!v.tiles) {
    // no tile maps (index data built ON THE DEVICE by mdq_env_topology's large-mesh instance, which emits the dof <-
    // element-slot lists but no tile positions): the element results go to the slab's element scratch (6 double2 per
    // triangle, free during the solves) and every row sums its slots in ascending order - the gather the right-hand sides use
    double2* es2 = reinterpret_cast<double2*>(v.work);
    for (int e = tid; e < v.nt; e += WG) {
      double2 xe[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) xe[i] = gx[v.cell_dofs[i * v.NT + e]];
      const Geo g = load_geo(v, e);
      double2 ye[6];
      op(e, g, (int)v.cell_outflow[e], xe, ye);
#pragma unroll
      for (int i = 0; i < 6; ++i) es2[e * 6 + i] = ye[i];
    }
    __syncthreads();
    for (int row = tid; row < n; row += WG) {
      double2 a = make_double2(0.0, 0.0);
      for (int s = v.g2_ptr[row]; s < v.g2_ptr[row + 1]; ++s) {
        const double2 c = es2[v.g2_src[s]];
        a.x += c.x;
        a.y += c.y;
      }
      epi(row, a.x, a.y);
    }
    __syncthreads();
    return;
  }
  tile_chunks_global(v, packed, es, ytmp, gx, op, 0, 1, true);
#ifdef MDQ_T5_TRACE
  long long t5q_ = __builtin_amdgcn_s_memtime();
#endif
  // epilogue: the sums of a batch of own rows are requested together (the caller's own loads follow row by row)
#ifndef MDQ_T5_EB
#define MDQ_T5_EB 8
#endif
  constexpr int EB = MDQ_T5_EB;
  for (int row0 = tid; row0 < n; row0 += EB * WG) {
    double2 a[EB];
#pragma unroll
    for (int k = 0; k < EB; ++k) a[k] = ytmp[min(row0 + k * WG, n - 1)];
#pragma unroll
    for (int k = 0; k < EB; ++k)
      if (row0 + k * WG < n) epi(row0 + k * WG, a[k].x, a[k].y);
  }
  T5_STAMP(4)
}

// ================================================================== sparse kernels (workgroup-wide)
//
// SELL-64, one thread per row: wave w owns slices w, w+NWAVE, ... (rows tid, tid+WG, ...), the
// same ownership as every `for (i = tid; i < n; i += WG)` vector pass, so a thread only ever
// reads/writes its own entries of the Krylov vectors; only the SpMV input vector is gathered.
// Matrix loads are perfectly coalesced (64 lanes x 32 B / 8 B contiguous per instruction).

template <class Epi>
__device__ __forceinline__ void spmv_sell_b2(const int32_t* __restrict__ sl_off, const int32_t* __restrict__ sl_col,
                                             const double* __restrict__ A, const double2* x, int n, Epi epi) {
  const int lane = threadIdx.x & 63;
  const int nsl = (n + 63) >> 6;
  const double4* __restrict__ A4 = reinterpret_cast<const double4*>(A);
  for (int s = threadIdx.x >> 6; s < nsl; s += NWAVE) {
    const int base = sl_off[s], w = (sl_off[s + 1] - base) >> 6;
    const double4* a = A4 + base + lane;
    const int32_t* c = sl_col + base + lane;
    double y0 = 0.0, y1 = 0.0;
#ifndef MDQ_SPMV_UNROLL
#define MDQ_SPMV_UNROLL 4
#endif
#pragma unroll MDQ_SPMV_UNROLL
    for (int j = 0; j < w; ++j) {
      const double4 av = a[j * 64];
      const double2 xv = x[c[j * 64]];
      y0 += av.x * xv.x + av.y * xv.y;
      y1 += av.z * xv.x + av.w * xv.y;
    }
    const int row = (s << 6) + lane;
    if (row < n) epi(row, y0, y1);
  }
}

template <class Epi>
__device__ __forceinline__ void spmv_sell_2rhs(const int32_t* __restrict__ sl_off, const int32_t* __restrict__ sl_col,
                                               const double* __restrict__ A, const double2* x, int n, Epi epi) {
  const int lane = threadIdx.x & 63;
  const int nsl = (n + 63) >> 6;
  for (int s = threadIdx.x >> 6; s < nsl; s += NWAVE) {
    const int base = sl_off[s], w = (sl_off[s + 1] - base) >> 6;
    const double* a = A + base + lane;
    const int32_t* c = sl_col + base + lane;
    double y0 = 0.0, y1 = 0.0;
#pragma unroll 4
    for (int j = 0; j < w; ++j) {
      const double av = a[j * 64];
      const double2 xv = x[c[j * 64]];
      y0 += av * xv.x;
      y1 += av * xv.y;
    }
    const int row = (s << 6) + lane;
    if (row < n) epi(row, y0, y1);
  }
}

template <class Epi>
__device__ __forceinline__ void spmv_sell(const int32_t* sl_off, const int32_t* sl_col, const double* A,
                                          const double* x, int n, Epi epi) {
  const int lane = threadIdx.x & 63;
  const int nsl = (n + 63) >> 6;
  for (int s = threadIdx.x >> 6; s < nsl; s += NWAVE) {
    const int base = sl_off[s], w = (sl_off[s + 1] - base) >> 6;
    const double* a = A + base + lane;
    const int32_t* c = sl_col + base + lane;
    double y0 = 0.0;
#pragma unroll 3
    for (int j = 0; j < w; ++j) y0 += a[j * 64] * x[c[j * 64]];
    const int row = (s << 6) + lane;
    if (row < n) epi(row, y0);
  }
}

// ================================================================== Krylov solvers
//
// MODE 0: assembled SELL operators, gather vectors in global memory (any mesh size)
// MODE 1: assembled SELL operators, gather vectors p / r resident in LDS
// MODE 2: matrix-free element-tile operators (x staged in LDS, tile of element results in LDS)

struct VelCtx {
  double2* x;    // solution (own rows)
  double2* r;    // residual / s (own rows)
  double2* rh;   // shadow residual (own rows)
  double2* p;    // search direction (own rows)
  double2* vv;   // A p (own rows)
  double2* t;    // A s (own rows)
  double2* gp;   // where SpMV #1 gathers p from   (MODE 2: the LDS stage buffer)
  double2* gr;   // where SpMV #2 gathers s from   (MODE 2: the same LDS stage buffer)
  double2* es;   // MODE 2 / 5: LDS element tile
  double2* yt;   // MODE 5: accumulation vector (global, own rows)
  bool packed;   // MODE 5: mf_scat holds packed words
  double a, mu;
};

// y = (D^-1 A1_bc) x on vectors that vanish on constrained dofs (see bicgstab_velocity)
__device__ __forceinline__ void velocity_op(const EnvView& v, double a, double mu, int e, int ko, const Geo& g,
                                            const double2 (&xe)[6], double2 (&ye)[6]);

template <int MODE, class Epi>
__device__ __forceinline__ void apply_velocity(const EnvView& v, const VelCtx& c, const double2* gx, Epi epi) {
  static_assert(MODE != 2, "the LDS-resident matrix-free mode has its own kernel (evolve_mf_kernel)");
  if constexpr (MODE == 5) {
    // full element operator on the gathered vector, rows scaled by D^-1 (0 on constrained rows) on the way out
    tile_apply_global(
        v, c.packed, c.es, c.yt, gx,
        [&](int e, const Geo& g, int ko, const double2(&xe)[6], double2(&ye)[6]) { velocity_op(v, c.a, c.mu, e, ko, g, xe, ye); },
        [&](int row, double y0, double y1) {
          const bool fl = v.bcu_flag[row] != 0;
          const double2 id = v.idiag1[row];
          epi(row, fl ? 0.0 : y0 * id.x, fl ? 0.0 : y1 * id.y);
        });
  } else {
    spmv_sell_b2(v.sl2_off, v.sl2_col, v.A1, gx, v.n2, epi);
  }
}

// BiCGStab on the row-scaled velocity system  (D^-1 A1_bc) x = D^-1 b.
// Entry: c.r holds the initial residual r0 = D^-1 (b - A x0) (zero on constrained dofs because x0
// already satisfies the Dirichlet values), bb = |D^-1 b|^2 and rr0 = |r0|^2 are workgroup-uniform.
// Every Krylov vector therefore vanishes on constrained dofs and the eliminated operator equals the
// plain element operator on them.  Exit: c.x holds the solution.
template <int MODE>
__device__ inline int bicgstab_velocity(const EnvView& v, const VelCtx& c, double rtol, int maxit, double bb,
                                        double rr0, double* red) {
  const int n = v.n2, tid = threadIdx.x;
  const double tol2 = rtol * rtol * bb;
  double rr = rr0;
  if (!(rr > tol2) || bb == 0.0) return 0;
  double rho = rr, rho_old = 1.0, alpha = 1.0, omega = 1.0;  // rh = r0, p = v = 0 on entry
  int it = 0;
  while (it < maxit) {
    ++it;
    const double beta = (rho / rho_old) * (alpha / omega);
    for (int i = tid; i < n; i += WG) {
      const double2 ri = c.r[i], pi = c.p[i], vi = c.vv[i];
      const double2 pn = make_double2(ri.x + beta * (pi.x - omega * vi.x), ri.y + beta * (pi.y - omega * vi.y));
      c.p[i] = pn;
      if (MODE == 2) c.gp[i] = pn;     // (mode 5: gp IS p)
    }
    __syncthreads();
    double a1[1] = {0.0};
    apply_velocity<MODE>(v, c, c.gp, [&](int row, double y0, double y1) {
      c.vv[row] = make_double2(y0, y1);
      const double2 h = c.rh[row];
      a1[0] += h.x * y0 + h.y * y1;
    });
    block_sum<1>(a1, red);
    if (a1[0] == 0.0) break;  // breakdown
    alpha = rho / a1[0];
    double a2[1] = {0.0};
    for (int i = tid; i < n; i += WG) {
      const double2 ri = c.r[i], vi = c.vv[i];
      const double2 sv = make_double2(ri.x - alpha * vi.x, ri.y - alpha * vi.y);
      c.r[i] = sv;
      if (MODE == 2) c.gr[i] = sv;
      a2[0] += sv.x * sv.x + sv.y * sv.y;
    }
    block_sum<1>(a2, red);  // (its barriers also publish s)
    if (!(a2[0] > tol2)) {
      for (int i = tid; i < n; i += WG) {
        const double2 xi = c.x[i], pi = c.p[i];
        c.x[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
      }
      break;
    }
    double a3[2] = {0.0, 0.0};
    apply_velocity<MODE>(v, c, c.gr, [&](int row, double y0, double y1) {
      c.t[row] = make_double2(y0, y1);
      const double2 sv = c.r[row];
      a3[0] += y0 * sv.x + y1 * sv.y;
      a3[1] += y0 * y0 + y1 * y1;
    });
    block_sum<2>(a3, red);
    if (a3[1] == 0.0) break;
    omega = a3[0] / a3[1];
    double a4[2] = {0.0, 0.0};
    for (int i = tid; i < n; i += WG) {
      const double2 ti = c.t[i], xi = c.x[i], pi = c.p[i], si = c.r[i], hi = c.rh[i];
      c.x[i] = make_double2(xi.x + alpha * pi.x + omega * si.x, xi.y + alpha * pi.y + omega * si.y);
      const double2 rn = make_double2(si.x - omega * ti.x, si.y - omega * ti.y);
      c.r[i] = rn;
      a4[0] += rn.x * rn.x + rn.y * rn.y;
      a4[1] += hi.x * rn.x + hi.y * rn.y;
    }
    block_sum<2>(a4, red);
    rr = a4[0];
    if (!(rr > tol2)) break;
    rho_old = rho;
    rho = a4[1];
    if (rho == 0.0 || omega == 0.0) break;
  }
  __syncthreads();
  return it;
}

struct MassCtx {
  double2* x;   // S x (own rows)
  double2* r;   // residual (own rows)
  double2* p;   // search direction (own rows)
  double2* q;   // M p (own rows)
  double2* gp;  // where the SpMV gathers p from
  double2* es;  // MODE 2 / 5: LDS element tile
  double2* yt;  // MODE 5: accumulation vector (global, own rows)
  bool packed;
};

// y = (S^-1 M_bc S^-1) x on vectors that vanish on constrained dofs
template <int MODE, class Epi>
__device__ __forceinline__ void apply_mass(const EnvView& v, const MassCtx& c, const double2* gx, Epi epi) {
  static_assert(MODE != 2, "the LDS-resident matrix-free mode has its own kernel (evolve_mf_kernel)");
  if constexpr (MODE == 5) {
    // gx holds S^-1 p (staged by the caller); rows scaled by S^-1 (0 on constrained rows) on the way out
    tile_apply_global(
        v, c.packed, c.es, c.yt, gx,
        [&](int, const Geo& g, int, const double2(&xe)[6], double2(&ye)[6]) { elem_mass(g, xe, ye); },
        [&](int row, double y0, double y1) {
          const double is = v.bcu_flag[row] ? 0.0 : 1.0 / v.sdiagM[row];
          epi(row, y0 * is, y1 * is);
        });
  } else {
    spmv_sell_2rhs(v.sl2_off, v.sl2_col, v.Ms, gx, v.n2, epi);
  }
}

// CG on the symmetrically scaled mass system, both velocity components at once.
// Entry: c.r = r0 = S^-1 (b - M S^-1 x0) with zeros on constrained dofs, c.p = r0 (and its staged copy).
template <int MODE>
__device__ inline int cg_mass(const EnvView& v, const MassCtx& c, double rtol, int maxit, double bb, double rr0,
                              double* red) {
  const int n = v.n2, tid = threadIdx.x;
  const double tol2 = rtol * rtol * bb;
  double rr = rr0;
  if (!(rr > tol2) || bb == 0.0) return 0;
  int it = 0;
  while (it < maxit) {
    ++it;
    double a1[1] = {0.0};
    apply_mass<MODE>(v, c, c.gp, [&](int row, double y0, double y1) {
      c.q[row] = make_double2(y0, y1);
      const double2 pi = c.p[row];
      a1[0] += pi.x * y0 + pi.y * y1;
    });
    block_sum<1>(a1, red);
    if (!(a1[0] > 0.0)) break;
    const double alpha = rr / a1[0];
    double a2[1] = {0.0};
    for (int i = tid; i < n; i += WG) {
      const double2 xi = c.x[i], pi = c.p[i], ri = c.r[i], qi = c.q[i];
      c.x[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
      const double2 rn = make_double2(ri.x - alpha * qi.x, ri.y - alpha * qi.y);
      c.r[i] = rn;
      a2[0] += rn.x * rn.x + rn.y * rn.y;
    }
    block_sum<1>(a2, red);
    const double rr_new = a2[0];
    if (!(rr_new > tol2)) break;
    const double beta = rr_new / rr;
    rr = rr_new;
    for (int i = tid; i < n; i += WG) {
      const double2 ri = c.r[i], pi = c.p[i];
      const double2 pn = make_double2(ri.x + beta * pi.x, ri.y + beta * pi.y);
      c.p[i] = pn;
      if (MODE == 2 || MODE == 5) {
        const double is = 1.0 / v.sdiagM[i];
        c.gp[i] = make_double2(pn.x * is, pn.y * is);
      }
    }
    __syncthreads();
  }
  __syncthreads();
  return it;
}

// CG on the symmetrically scaled pressure system; vectors (and, when it fits, the matrix) in LDS.
__device__ inline int cg_pressure(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, double rtol,
                                  int maxit, double* x, double* r, double* p, double* q, double* red) {
  const int tid = threadIdx.x;
  double acc[2] = {0.0, 0.0};
  __syncthreads();
  spmv_sell(sl_off, sl_col, A, x, n, [&](int row, double y0) {
    const double b = r[row];
    const double rr = b - y0;
    r[row] = rr;
    p[row] = rr;
    acc[0] += b * b;
    acc[1] += rr * rr;
  });
  block_sum<2>(acc, red);
  const double bb = acc[0];
  double rr = acc[1];
  const double tol2 = rtol * rtol * bb;
  if (!(rr > tol2) || bb == 0.0) return 0;
  int it = 0;
  while (it < maxit) {
    ++it;
    double a1[1] = {0.0};
    spmv_sell(sl_off, sl_col, A, p, n, [&](int row, double y0) {
      q[row] = y0;
      a1[0] += p[row] * y0;
    });
    block_sum<1>(a1, red);
    if (!(a1[0] > 0.0)) break;
    const double alpha = rr / a1[0];
    double a2[1] = {0.0};
    for (int i = tid; i < n; i += WG) {
      x[i] += alpha * p[i];
      const double rn = r[i] - alpha * q[i];
      r[i] = rn;
      a2[0] += rn * rn;
    }
    block_sum<1>(a2, red);
    const double rr_new = a2[0];
    if (!(rr_new > tol2)) break;
    const double beta = rr_new / rr;
    rr = rr_new;
    for (int i = tid; i < n; i += WG) p[i] = r[i] + beta * p[i];
    __syncthreads();
  }
  __syncthreads();
  return it;
}

// Jacobi-scaled CG on the SELL-64 P1 Laplacian for the three-kernel mode 3: every thread keeps x, r, q of its own
// two rows (slice = wave, wave + NTH/64; lane = row in the slice: the rows its SpMV produces) in registers, only the
// search direction lives in LDS for the gather; one-barrier reductions.  3 barriers per iteration instead of 6.
// Needs n <= 2 * NTH (the caller falls back to cg_pressure otherwise).  x, r, p: LDS vectors as in cg_pressure.
template <int NTH>
__device__ inline int cg_pressure_reg(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, double rtol,
                                      int maxit, double* x, double* r, double* p, double* red, int& rsel) {
  constexpr int NW = NTH / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsl = (n + 63) >> 6;
  int base[2], wid[2], row[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s_ = wave + NW * k;
    row[k] = (s_ << 6) + lane;
    base[k] = s_ < nsl ? sl_off[s_] : 0;
    wid[k] = s_ < nsl ? (sl_off[s_ + 1] - base[k]) >> 6 : 0;
  }
  auto spmv = [&](const double* vec, double(&y)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double* a = A + base[k] + lane;
      const int32_t* c = sl_col + base[k] + lane;
      double y0 = 0.0;
#pragma unroll 4
      for (int j = 0; j < wid[k]; ++j) y0 += a[j * 64] * vec[c[j * 64]];
      y[k] = y0;
    }
  };
  __syncthreads();
  double y[2], xv[2], rv[2], pv[2];
  spmv(x, y);
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    xv[k] = rv[k] = pv[k] = 0.0;
    if (row[k] < n) {
      const double b = r[row[k]];
      xv[k] = x[row[k]];
      rv[k] = pv[k] = b - y[k];
      p[row[k]] = pv[k];
      acc[0] += b * b;
      acc[1] += rv[k] * rv[k];
    }
  }
  block_sum<2, NW>(acc, red);
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  int it = 0;
  if (rr > tol2 && bb != 0.0) {
    while (it < maxit) {
      ++it;
      double q[2];
      spmv(p, q);
      double a1[1] = {pv[0] * q[0] + pv[1] * q[1]};
      block_sum1<1, NW>(a1, red, rsel);
      if (!(a1[0] > 0.0)) break;
      const double alpha = rr / a1[0];
      double a2[1] = {0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xv[k] += alpha * pv[k];
        rv[k] -= alpha * q[k];
        a2[0] += rv[k] * rv[k];
      }
      block_sum1<1, NW>(a2, red, rsel);
      const double rr_new = a2[0];
      if (!(rr_new > tol2)) break;
      const double beta = rr_new / rr;
      rr = rr_new;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pv[k] = rv[k] + beta * pv[k];
        if (row[k] < n) p[row[k]] = pv[k];
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (row[k] < n) x[row[k]] = xv[k];
  __syncthreads();
  return it;
}

// cg_pressure_reg with the matrix rows of a thread in REGISTERS and a BRANCH-FREE operator application.  Counters of the
// LDS-resident version (profiles/r03_sq_summary.json): LDS utilisation 20 %, bank conflicts 0.3 %, VALU busy 18 % - the
// iteration is a chain of LDS LATENCIES, not bandwidth: per row three batches of (index, value) reads followed by the
// dependent gathers, two rows one after the other, ~2 100 of the iteration's ~3 400 cycles.  Here a thread loads the
// up to RW entries of its two rows once (slots past a slice's width: value 0, column = own row, like the SELL padding),
// and an application issues all 2 x RW gathers back to back - no branch, so nothing waits between them - and then runs
// the two fma chains in the same column order as before (adding 0 * p[row] for the padded slots).  (The Chebyshev
// variant below keeps its rows in registers too but guards every gather with a branch on the slice width, which
// serialises the round trips: that is why it never beat the LDS version.)  Needs n <= 2 * NTH and slices of at most RW
// entries per row: returns -1 otherwise (the caller takes cg_pressure_reg).
template <int NTH, int RW = 16>
__device__ inline int cg_pressure_regm(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, double rtol,
                                       int maxit, double* x, double* r, double* p, double* red, int& rsel) {
  constexpr int NW = NTH / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsl = (n + 63) >> 6;
  int row[2];
  double av[2][RW];
  int co[2][RW];                      // byte offsets into the gathered vector
  bool fits = true;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s_ = wave + NW * k;
    row[k] = (s_ << 6) + lane;
    const int base = s_ < nsl ? sl_off[s_] : 0;
    const int wid = s_ < nsl ? (sl_off[s_ + 1] - base) >> 6 : 0;
    fits = fits && wid <= RW;
    const int self = min(row[k], n - 1);
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const bool in = j < wid;
      av[k][j] = in ? A[base + lane + j * 64] : 0.0;
      co[k][j] = 8 * (in ? sl_col[base + lane + j * 64] : self);
    }
  }
  // workgroup-wide OR of `!fits` through the (still unused) search-direction vector (no static LDS in this kernel)
  if (threadIdx.x == 0) p[NW] = 0.0;
  __syncthreads();
  if (!fits) p[NW] = 1.0;
  __syncthreads();
  const bool toowide = p[NW] != 0.0;
  __syncthreads();
  if (toowide) return -1;
  auto spmv = [&](const double* vec, double(&y)[2]) {
    const char* vb = reinterpret_cast<const char*>(vec);
    double g[2][RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      g[0][j] = *reinterpret_cast<const double*>(vb + co[0][j]);
      g[1][j] = *reinterpret_cast<const double*>(vb + co[1][j]);
    }
    double y0 = 0.0, y1 = 0.0;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      y0 += av[0][j] * g[0][j];
      y1 += av[1][j] * g[1][j];
    }
    y[0] = y0;
    y[1] = y1;
  };
  __syncthreads();
  double y[2], xv[2], rv[2], pv[2];
  spmv(x, y);
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    xv[k] = rv[k] = pv[k] = 0.0;
    if (row[k] < n) {
      const double b = r[row[k]];
      xv[k] = x[row[k]];
      rv[k] = pv[k] = b - y[k];
      p[row[k]] = pv[k];
      acc[0] += b * b;
      acc[1] += rv[k] * rv[k];
    }
  }
  block_sum<2, NW>(acc, red);
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  int it = 0;
  if (rr > tol2 && bb != 0.0) {
    while (it < maxit) {
      ++it;
      double q[2];
      spmv(p, q);
      double a1[1] = {pv[0] * q[0] + pv[1] * q[1]};
      block_sum1<1, NW>(a1, red, rsel);
      if (!(a1[0] > 0.0)) break;
      const double alpha = rr / a1[0];
      double a2[1] = {0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xv[k] += alpha * pv[k];
        rv[k] -= alpha * q[k];
        a2[0] += rv[k] * rv[k];
      }
      block_sum1<1, NW>(a2, red, rsel);
      const double rr_new = a2[0];
      if (!(rr_new > tol2)) break;
      const double beta = rr_new / rr;
      rr = rr_new;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pv[k] = rv[k] + beta * pv[k];
        if (row[k] < n) p[row[k]] = pv[k];
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (row[k] < n) x[row[k]] = xv[k];
  __syncthreads();
  return it;
}

// The same CG with the matrix rows of a thread in REGISTERS (values + column offsets of its two rows: they do not change
// over the iterations, and the LDS-resident version paid two dependent LDS round trips per entry - column, then vector
// element) and, for degree > 1, a CHEBYSHEV polynomial preconditioner on top of the Jacobi scaling: z = p_m(A) r with
// p_m the degree-(m - 1) Chebyshev approximation of 1 / lambda on [lmax / CHEB_RATIO, lmax] (lmax: Gershgorin bound of the
// scaled matrix, computed here).  p_m(A) is symmetric positive definite for any spectrum inside (0, lmax], so this is a
// plain preconditioned CG; it trades reductions (two workgroup barriers each) for operator applications (one barrier):
// m - 1 extra applications per iteration, roughly m times fewer iterations.  Needs n <= 2 * NTH and SELL slices of at
// most PCG_W entries per row (else: returns -1 and the caller takes cg_pressure_reg).
constexpr int PCG_W = 12;
constexpr double CHEB_RATIO = 30.0;
template <int NTH>
__device__ inline int cg_pressure_cheb(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, double rtol,
                                       int maxit, int degree, double* x, double* r, double* p, double* red, int& rsel) {
  constexpr int NW = NTH / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsl = (n + 63) >> 6;
  int row[2], wid[2];
  double av[2][PCG_W];
  int cv[2][PCG_W];
  bool fits = true;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s_ = wave + NW * k;
    row[k] = (s_ << 6) + lane;
    const int base = s_ < nsl ? sl_off[s_] : 0;
    wid[k] = s_ < nsl ? (sl_off[s_ + 1] - base) >> 6 : 0;
    fits = fits && wid[k] <= PCG_W;
#pragma unroll
    for (int j = 0; j < PCG_W; ++j) {
      const bool in = j < wid[k];
      av[k][j] = in ? A[base + lane + j * 64] : 0.0;
      cv[k][j] = in ? sl_col[base + lane + j * 64] : 0;
    }
  }
  // workgroup-wide OR of `!fits` through the (still unused) search-direction vector: __syncthreads_or would add a static
  // LDS word to kernels that already ask for the whole 160 KB dynamically
  if (threadIdx.x == 0) p[NW] = 0.0;
  __syncthreads();
  if (!fits) p[NW] = 1.0;
  __syncthreads();
  const bool toowide = p[NW] != 0.0;
  __syncthreads();
  if (toowide) return -1;
  auto spmv = [&](const double* vec, double(&y)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double y0 = 0.0, y1 = 0.0;
#pragma unroll
      for (int j = 0; j < PCG_W; j += 2) {
        if (j < wid[k]) y0 += av[k][j] * vec[cv[k][j]];            // (wid is uniform over the wave: scalar branches)
        if (j + 1 < wid[k]) y1 += av[k][j + 1] * vec[cv[k][j + 1]];
      }
      y[k] = y0 + y1;
    }
  };
  // Gershgorin bound of the largest eigenvalue (the scaled matrix has a unit diagonal; rows of constrained vertices are
  // identity rows)
  double lmax = 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double s_ = 0.0;
#pragma unroll
    for (int j = 0; j < PCG_W; ++j) s_ += fabs(av[k][j]);
    lmax = fmax(lmax, row[k] < n ? s_ : 0.0);
  }
  for (int off = 32; off > 0; off >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, off));
  {
    double* buf = p;                          // (the search direction is not in use yet)
    if (lane == 0) buf[wave] = lmax;
    __syncthreads();
    double m_ = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) m_ = fmax(m_, buf[i]);
    lmax = m_;
    __syncthreads();
  }
  const int M = degree < 1 ? 1 : degree;
  const double lmin = lmax / CHEB_RATIO, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  double y[2], xv[2], rv[2], pv[2], zv[2];
  spmv(x, y);
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    xv[k] = rv[k] = pv[k] = zv[k] = 0.0;
    if (row[k] < n) {
      const double b = r[row[k]];
      xv[k] = x[row[k]];
      rv[k] = b - y[k];
      acc[0] += b * b;
      acc[1] += rv[k] * rv[k];
    }
  }
  block_sum<2, NW>(acc, red);
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  int it = 0;
  if (rr > tol2 && bb != 0.0) {
    double rz_old = 1.0;
    // `r` (LDS) is free from here on: the polynomial's iterates are staged there for the gathers
    double* zs = r;
    while (it < maxit) {
      // z = p_M(A) r  (Chebyshev iteration for A z = r from z = 0, M - 1 further applications)
      double dv[2];
      double rho = 1.0 / sigma;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        dv[k] = rv[k] / theta;
        zv[k] = dv[k];
      }
      for (int m_ = 1; m_ < M; ++m_) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (row[k] < n) zs[row[k]] = zv[k];
        __syncthreads();
        double az[2];
        spmv(zs, az);
        const double rho_n = 1.0 / (2.0 * sigma - rho);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          dv[k] = rho_n * rho * dv[k] + (2.0 * rho_n / delta) * (rv[k] - az[k]);
          zv[k] += dv[k];
        }
        rho = rho_n;
        __syncthreads();                         // (everybody has gathered from zs before the next round rewrites it)
      }
      double a0[1] = {rv[0] * zv[0] + rv[1] * zv[1]};
      block_sum1<1, NW>(a0, red, rsel);
      const double rz = a0[0];
      const double beta = it == 0 ? 0.0 : rz / rz_old;
      rz_old = rz;
      ++it;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pv[k] = zv[k] + beta * pv[k];
        if (row[k] < n) p[row[k]] = pv[k];
      }
      __syncthreads();
      double q[2];
      spmv(p, q);
      double a1[1] = {pv[0] * q[0] + pv[1] * q[1]};
      block_sum1<1, NW>(a1, red, rsel);
      if (!(a1[0] > 0.0)) break;
      const double alpha = rz / a1[0];
      double a2[1] = {0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xv[k] += alpha * pv[k];
        rv[k] -= alpha * q[k];
        a2[0] += rv[k] * rv[k];
      }
      block_sum1<1, NW>(a2, red, rsel);
      rr = a2[0];
      if (!(rr > tol2)) break;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (row[k] < n) x[row[k]] = xv[k];
  __syncthreads();
  return it;
}

// The register-resident CG (cg_pressure_reg) with a TWO-LEVEL additive preconditioner on top of the Jacobi scaling:
//      z = r + P A_c^-1 P^T r,     A_c = P^T A P,
// P = piecewise constants over NAGX x NAGY geometric aggregates: vertices ranked by x into NAGX strips of equal
// population, every strip ranked by y into NAGY cells (compact patches of ~16 vertices; aggregates of consecutive INDICES
// do nothing for this matrix: tools + HISTORY 8.1).  A_c (56 x 56) is accumulated with LDS atomics, inverted in LDS by
// Gauss-Jordan (SPD: no pivoting) once per solve and kept in fp32 (a preconditioner may be approximate; it is stored
// exactly symmetric).  Per iteration: per-wave partial restrictions by LDS atomics (rows of one aggregate are scattered
// over the waves), one 56^2 product spread over 448 threads, prolongation by one LDS read per row: two extra barriers.
// Iterations on ys930 (developed flow): 154 -> 86, same answers.  MEASURED (tools/time_pcg.py): 465 us per solve against 247 us of
// the plain Jacobi-CG - the O(n^2) rank counts of the aggregation and the 56-step inversion cost ~0.1 ms per solve, and
// every iteration pays two more barriers, the atomics and a 63-read coarse product per thread.  An O(n) aggregation
// (histograms) and one shared restriction vector would bring it to ~200 us (-18 %): kept as an OPTION (pcg_degree < 0), the
// default stays the Jacobi-CG.  LDS: the four pressure vectors' space (x, r, p, q = 4 NVp doubles, NVp >= 1024):
// x and r live in registers during the iterations.  Needs n <= 2 NTH, n <= 1024 and at least NAG vertices; else -1.
constexpr int NAGX = 8, NAGY = 7, NAG = NAGX * NAGY;
template <int NTH>
__device__ inline int cg_pressure_2l(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, const double* coords,
                                     double rtol, int maxit, double* x, double* r, double* p, int NVp, double* red, int& rsel) {
  constexpr int NW = NTH / 64;
  if (n > 2 * NTH || n > 1024 || n < 4 * NAG || NVp < 1024) return -1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsl = (n + 63) >> 6;
  int base[2], wid[2], row[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s_ = wave + NW * k;
    row[k] = (s_ << 6) + lane;
    base[k] = s_ < nsl ? sl_off[s_] : 0;
    wid[k] = s_ < nsl ? (sl_off[s_ + 1] - base[k]) >> 6 : 0;
  }
  auto spmv = [&](const double* vec, double(&y)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double* a = A + base[k] + lane;
      const int32_t* c = sl_col + base[k] + lane;
      double y0 = 0.0;
#pragma unroll 4
      for (int j = 0; j < wid[k]; ++j) y0 += a[j * 64] * vec[c[j * 64]];
      y[k] = y0;
    }
  };
  // LDS carve-up of the 4 NVp doubles behind x (x | r | p | q are contiguous in every caller)
  double* W = x;
  double* AC = W;                                            // [NAG][NAG] fp64 while it is built and inverted
  float* AI = reinterpret_cast<float*>(W);                   // ... then fp32 in its first half
  double* YV = W + NAG * NAG / 2 + 8;                        // [NAG] coarse result       (behind the fp32 inverse)
  double* PG = W + 2 * NVp;                                  // the gather copy of the search direction (= p)
  double* WP = W + 3 * NVp;                                  // [NW][NAG] per-wave partial restrictions
  unsigned char* AG = reinterpret_cast<unsigned char*>(W + 3 * NVp + NW * NAG);   // [n] aggregate of a vertex
  static_assert(NAG * NAG <= 3 * 1024 + 64 && 16 * NAG + 128 <= 1024, "two-level scratch");
  __syncthreads();
  double y[2], xv[2], rv[2], pv[2], zv[2];
  spmv(x, y);
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    xv[k] = rv[k] = pv[k] = zv[k] = 0.0;
    if (row[k] < n) {
      const double b = r[row[k]];
      xv[k] = x[row[k]];
      rv[k] = b - y[k];
      acc[0] += b * b;
      acc[1] += rv[k] * rv[k];
    }
  }
  block_sum<2, NW>(acc, red);                                // (its barriers: x and r are in registers, their LDS space is free)
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  int it = 0;
  if (rr > tol2 && bb != 0.0) {
    // ---- aggregates: rank by x -> strip, rank by y inside the strip -> cell
    double* XC = W;                                          // coordinates staged in LDS for the counting loops
    double* YC = W + 1024;
    unsigned char* ST = reinterpret_cast<unsigned char*>(W + 2048);
    for (int i = tid; i < n; i += NTH) {
      XC[i] = coords[2 * i];
      YC[i] = coords[2 * i + 1];
    }
    __syncthreads();
    int strip[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (row[k] < n) {
        const double xi = XC[row[k]];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
          const double xj = XC[j];
          rank += (xj < xi) | ((xj == xi) & (j < row[k]));
        }
        strip[k] = rank * NAGX / n;
        ST[row[k]] = (unsigned char)strip[k];
      }
    __syncthreads();
    int agg[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (row[k] < n) {
        const double yi = YC[row[k]];
        int rank = 0, cnt = 0;
        for (int j = 0; j < n; ++j) {
          const bool same = ST[j] == strip[k];
          const double yj = YC[j];
          cnt += same;
          rank += same & ((yj < yi) | ((yj == yi) & (j < row[k])));
        }
        agg[k] = strip[k] * NAGY + rank * NAGY / cnt;
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (row[k] < n) AG[row[k]] = (unsigned char)agg[k];
    for (int e = tid; e < NAG * NAG; e += NTH) AC[e] = 0.0;
    __syncthreads();
    // ---- coarse matrix
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (row[k] < n) {
        const double* a = A + base[k] + lane;
        const int32_t* c = sl_col + base[k] + lane;
        for (int j = 0; j < wid[k]; ++j) {
          const double av = a[j * 64];
          if (av != 0.0) atomicAdd(&AC[agg[k] * NAG + AG[c[j * 64]]], av);
        }
      }
    __syncthreads();
    // ---- in-place Gauss-Jordan inverse (every thread owns the elements e = tid + m NTH)
    constexpr int EPT = (NAG * NAG + NTH - 1) / NTH;
    int ei_[EPT], ej_[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      const int e = tid + m * NTH;
      ei_[m] = e / NAG;
      ej_[m] = e - ei_[m] * NAG;
    }
    for (int kk = 0; kk < NAG; ++kk) {
      const double ip = 1.0 / AC[kk * NAG + kk];
      double nv_[EPT];
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * NTH;
        nv_[m] = 0.0;
        if (e < NAG * NAG) {
          const int i = ei_[m], j = ej_[m];
          const double aij = AC[e], aik = AC[i * NAG + kk], akj = AC[kk * NAG + j];
          nv_[m] = i == kk ? (j == kk ? ip : akj * ip) : (j == kk ? -aik * ip : aij - aik * akj * ip);
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * NTH;
        if (e < NAG * NAG) AC[e] = nv_[m];
      }
      __syncthreads();
    }
    {  // fp32 copy, exactly symmetric: the entry of the upper triangle for both (i, j) and (j, i)
      float f_[EPT];
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * NTH;
        f_[m] = 0.f;
        if (e < NAG * NAG) {
          const int i = e / NAG, j = e - i * NAG;
          f_[m] = (float)AC[min(i, j) * NAG + max(i, j)];
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * NTH;
        if (e < NAG * NAG) AI[e] = f_[m];
      }
    }
    __syncthreads();
    double rz_old = 1.0;
    while (it < maxit) {
      // ---- z = r + P A_c^-1 P^T r
      if (lane < NAG) WP[wave * NAG + lane] = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (row[k] < n) atomicAdd(&WP[wave * NAG + agg[k]], rv[k]);
      __syncthreads();
      if (tid < NAG * 8) {
        const int I = tid >> 3, part = tid & 7;
        double s_ = 0.0;
        for (int J = part; J < NAG; J += 8) {
          double w_ = 0.0;
#pragma unroll
          for (int q = 0; q < NW; ++q) w_ += WP[q * NAG + J];
          s_ += (double)AI[I * NAG + J] * w_;
        }
        s_ += dpp_get<0xB1, 0xF>(s_);      // quad_perm [1,0,3,2]
        s_ += dpp_get<0x4E, 0xF>(s_);      // quad_perm [2,3,0,1]
        s_ += dpp_get<0x141, 0xF>(s_);     // row_half_mirror: the other quad of the group of 8
        if (part == 0) YV[I] = s_;
      }
      __syncthreads();
      double a0[1] = {0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        zv[k] = row[k] < n ? rv[k] + YV[agg[k]] : 0.0;
        a0[0] += rv[k] * zv[k];
      }
      block_sum1<1, NW>(a0, red, rsel);
      const double rz = a0[0];
      const double beta = it == 0 ? 0.0 : rz / rz_old;
      rz_old = rz;
      ++it;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        pv[k] = zv[k] + beta * pv[k];
        if (row[k] < n) PG[row[k]] = pv[k];
      }
      __syncthreads();
      double q[2];
      spmv(PG, q);
      double a1[1] = {pv[0] * q[0] + pv[1] * q[1]};
      block_sum1<1, NW>(a1, red, rsel);
      if (!(a1[0] > 0.0)) break;
      const double alpha = rz / a1[0];
      double a2[1] = {0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xv[k] += alpha * pv[k];
        rv[k] -= alpha * q[k];
        a2[0] += rv[k] * rv[k];
      }
      block_sum1<1, NW>(a2, red, rsel);
      rr = a2[0];
      if (!(rr > tol2)) break;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (row[k] < n) x[row[k]] = xv[k];
  __syncthreads();
  return it;
}

// The two-level preconditioner for the meshes BEYOND the register-resident solvers (cg_pressure: four LDS vectors, 512
// threads, any n up to the LDS): where Jacobi-CG needs hundreds of iterations (refined ys930, 3 322 vertices: 320-343).
// Same operator z = r + P A_c^-1 P^T r over NAGX x NAGY geometric aggregates as cg_pressure_2l, with what that one's header
// lists as missing: the aggregation in O(n) - a 1 024-bin histogram of x, its prefix sums cut into NAGX strips of (about)
// equal population, per strip a 256-bin histogram of y cut into NAGY cells; vertices of one fine bin share a strip / cell,
// an empty aggregate gets a unit diagonal - and one-barrier reductions.  Scratch: the histograms and then the fp64 coarse
// matrix in the p | q vectors (free until the first search direction), the fp32 inverse, the per-wave partial
// restrictions and the aggregate of every vertex in `extra` (LDS: the fifth pressure vector, the direct solver's scratch).
// The four CG vectors may as well live in GLOBAL memory (meshes beyond ~4 000 vertices, the PG instances of the kernels:
// `scratch` is then the search direction's slab vector, `extra` the LDS the pressure vectors do not occupy).
// Returns -1 where it does not apply (too few vertices, a scratch too small): the caller runs cg_pressure.
constexpr int TL_XB = 1024, TL_YB = 256;
__host__ __device__ inline size_t tl_extra_bytes(int n) {
  return sizeof(float) * NAG * NAG + sizeof(double) * ((WG / 64) * NAG + NAG) + (size_t)((n + 7) & ~7);
}
// The set-up of the two-level preconditioner (shared by cg_pressure_2l_lds and cg_pressure_2l_onchip): aggregates AG[n] by the
// histogram rule, coarse matrix by LDS atomics, its Gauss-Jordan inverse, the exactly symmetric fp32 copy AI [NAG][NAG].
// `hist`: LDS scratch of 4 (TL_XB + NAGX TL_YB + 2 NAGX) + n bytes; AC: LDS [NAG][NAG] doubles (may alias `hist`: the
// histograms are dead when it is zeroed); red: the reduction scratch ([32..63] used here).  Ends WITHOUT a barrier behind AI.
__device__ __forceinline__ void tl_build(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, const double* coords,
                                         int* hist, double* AC, float* AI, unsigned char* AG, double* red) {
  constexpr int NW = WG / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- aggregates
  {
    int* HX = hist;                               // [TL_XB] counts, then exclusive prefix sums
    int* HY = HX + TL_XB;                         // [NAGX][TL_YB]
    int* SC = HY + NAGX * TL_YB;                  // [NAGX] population of a strip ([NW] wave totals of the scan behind it)
    unsigned char* ST = reinterpret_cast<unsigned char*>(SC + 2 * NAGX);   // [n] strip of a vertex
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int i = tid; i < n; i += WG) {
      const double cx = coords[2 * i], cy = coords[2 * i + 1];
      lo[0] = fmin(lo[0], cx);
      hi[0] = fmax(hi[0], cx);
      lo[1] = fmin(lo[1], cy);
      hi[1] = fmax(hi[1], cy);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
      for (int off = 32; off > 0; off >>= 1) {
        lo[c] = fmin(lo[c], __shfl_xor(lo[c], off));
        hi[c] = fmax(hi[c], __shfl_xor(hi[c], off));
      }
    if (lane == 0) {
      red[32 + wave] = lo[0];
      red[40 + wave] = hi[0];
      red[48 + wave] = lo[1];
      red[56 + wave] = hi[1];
    }
    for (int k = tid; k < TL_XB + NAGX * TL_YB + NAGX; k += WG) HX[k] = 0;
    __syncthreads();
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) {
      lo[0] = fmin(lo[0], red[32 + w_]);
      hi[0] = fmax(hi[0], red[40 + w_]);
      lo[1] = fmin(lo[1], red[48 + w_]);
      hi[1] = fmax(hi[1], red[56 + w_]);
    }
    const double sx = hi[0] > lo[0] ? TL_XB / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? TL_YB / (hi[1] - lo[1]) : 0.0;
    auto xbin = [&](int i) { return min(TL_XB - 1, (int)((coords[2 * i] - lo[0]) * sx)); };
    auto ybin = [&](int i) { return min(TL_YB - 1, (int)((coords[2 * i + 1] - lo[1]) * sy)); };
    for (int i = tid; i < n; i += WG) atomicAdd(&HX[xbin(i)], 1);
    __syncthreads();
    {  // exclusive prefix sums of the 1 024 counts: two bins per thread, scan over the workgroup
      const int c0 = HX[2 * tid], c1 = HX[2 * tid + 1];
      int s_ = c0 + c1;
      for (int off = 1; off < 64; off <<= 1) {
        const int t_ = __shfl_up(s_, off);
        if (lane >= off) s_ += t_;
      }
      int* WT = SC + NAGX;                        // [NW] wave totals
      if (lane == 63) WT[wave] = s_;
      __syncthreads();
      int before = 0;
      for (int w_ = 0; w_ < wave; ++w_) before += WT[w_];
      const int ex = before + s_ - c0 - c1;
      HX[2 * tid] = ex;
      HX[2 * tid + 1] = ex + c0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += WG) {
      const int st = min(NAGX - 1, (int)((long long)HX[xbin(i)] * NAGX / n));
      ST[i] = (unsigned char)st;
      atomicAdd(&SC[st], 1);
      atomicAdd(&HY[st * TL_YB + ybin(i)], 1);
    }
    __syncthreads();
    {  // per strip (= wave): exclusive prefix sums of its 256 counts, four bins per lane
      int* h = HY + wave * TL_YB + 4 * lane;
      const int c0 = h[0], c1 = h[1], c2 = h[2], c3 = h[3];
      int s_ = c0 + c1 + c2 + c3;
      for (int off = 1; off < 64; off <<= 1) {
        const int t_ = __shfl_up(s_, off);
        if (lane >= off) s_ += t_;
      }
      const int ex = s_ - (c0 + c1 + c2 + c3);
      h[0] = ex;
      h[1] = ex + c0;
      h[2] = ex + c0 + c1;
      h[3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
    for (int i = tid; i < n; i += WG) {
      const int st = ST[i];
      const int cell = min(NAGY - 1, (int)((long long)HY[st * TL_YB + ybin(i)] * NAGY / max(SC[st], 1)));
      AG[i] = (unsigned char)(st * NAGY + cell);
    }
  }
  // ---- coarse matrix (fp64, in p | q), its inverse, the fp32 copy
  // (in LDS behind the tables when `extra` has the room - the instances with global vectors -, else in the scratch)
  __syncthreads();
  for (int e = tid; e < NAG * NAG; e += WG) AC[e] = 0.0;
  __syncthreads();
  {
    const int nsl = (n + 63) >> 6;
    for (int s_ = wave; s_ < nsl; s_ += NW) {
      const int base = sl_off[s_], wd = (sl_off[s_ + 1] - base) >> 6;
      const int row = (s_ << 6) + lane;
      if (row < n) {
        const int ar = AG[row] * NAG;
        for (int j = 0; j < wd; ++j) {
          const double av = A[base + lane + j * 64];
          if (av != 0.0) atomicAdd(&AC[ar + AG[sl_col[base + lane + j * 64]]], av);
        }
      }
    }
  }
  __syncthreads();
  if (tid < NAG && AC[tid * NAG + tid] == 0.0) AC[tid * NAG + tid] = 1.0;   // (an empty aggregate)
  __syncthreads();
  {
    constexpr int EPT = (NAG * NAG + WG - 1) / WG;
    int ei_[EPT], ej_[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      const int e = tid + m * WG;
      ei_[m] = e / NAG;
      ej_[m] = e - ei_[m] * NAG;
    }
    for (int kk = 0; kk < NAG; ++kk) {
      const double ip = 1.0 / AC[kk * NAG + kk];
      double nv_[EPT];
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * WG;
        nv_[m] = 0.0;
        if (e < NAG * NAG) {
          const int i = ei_[m], j = ej_[m];
          const double aij = AC[e], aik = AC[i * NAG + kk], akj = AC[kk * NAG + j];
          nv_[m] = i == kk ? (j == kk ? ip : akj * ip) : (j == kk ? -aik * ip : aij - aik * akj * ip);
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        const int e = tid + m * WG;
        if (e < NAG * NAG) AC[e] = nv_[m];
      }
      __syncthreads();
    }
    // fp32 copy, exactly symmetric: the entry of the upper triangle for both (i, j) and (j, i)
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
      const int e = tid + m * WG;
      if (e < NAG * NAG) AI[e] = (float)AC[min(ei_[m], ej_[m]) * NAG + max(ei_[m], ej_[m])];
    }
  }
}

__device__ __attribute__((noinline)) int cg_pressure_2l_lds(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A, const double* coords,
                                         double rtol, int maxit, double* x, double* r, double* p, double* q,
                                         size_t scratch_bytes, double* extra, size_t extra_bytes, double* red) {
  constexpr int NW = WG / 64;
  static_assert(NW == NAGX, "one wave per strip in the y pass");
  static_assert(sizeof(float) * NAG * NAG % 8 == 0, "alignment of the partial restrictions");
  // (`scratch_bytes`: what is contiguous behind p - p | q in LDS, the slab vector of p in global memory)
  if (n < 4 * NAG || tl_extra_bytes(n) > extra_bytes || scratch_bytes < sizeof(double) * NAG * NAG ||
      scratch_bytes < sizeof(int) * (TL_XB + NAGX * TL_YB + 2 * NAGX) + (size_t)n)
    return -1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* AI = reinterpret_cast<float*>(extra);                             // [NAG][NAG] inverse of the coarse matrix
  double* WP = extra + NAG * NAG / 2;                                      // [NW][NAG] per-wave partial restrictions
  double* YV = WP + NW * NAG;                                              // [NAG] coarse result
  unsigned char* AG = reinterpret_cast<unsigned char*>(YV + NAG);          // [n] aggregate of a vertex
  double acc[2] = {0.0, 0.0};
  __syncthreads();
  spmv_sell(sl_off, sl_col, A, x, n, [&](int row, double y0) {
    const double b = r[row];
    const double r0 = b - y0;
    r[row] = r0;
    acc[0] += b * b;
    acc[1] += r0 * r0;
  });
  block_sum<2>(acc, red);
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  if (!(rr > tol2) || bb == 0.0) return 0;
  // ---- aggregates, coarse matrix (fp64: in LDS behind the tables when `extra` has the room - the instances with global
  //      vectors -, else in the scratch p | q), its inverse, the fp32 copy
  tl_build(n, sl_off, sl_col, A, coords, reinterpret_cast<int*>(p),
           extra_bytes >= tl_extra_bytes(n) + sizeof(double) * NAG * NAG
               ? reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(extra) + tl_extra_bytes(n)) : p,
           AI, AG, red);
  __syncthreads();
  int rsel = 0;
  double rz_old = 1.0;
  int it = 0;
  while (it < maxit) {
    // ---- z = r + P A_c^-1 P^T r
    if (lane < NAG) WP[wave * NAG + lane] = 0.0;
    for (int i = tid; i < n; i += WG) atomicAdd(&WP[wave * NAG + AG[i]], r[i]);
    __syncthreads();
    if (tid < NAG * 8) {
      const int I = tid >> 3, part = tid & 7;
      double s_ = 0.0;
      for (int J = part; J < NAG; J += 8) {
        double w_ = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) w_ += WP[w2 * NAG + J];
        s_ += (double)AI[I * NAG + J] * w_;
      }
      s_ += dpp_get<0xB1, 0xF>(s_);      // quad_perm [1,0,3,2]
      s_ += dpp_get<0x4E, 0xF>(s_);      // quad_perm [2,3,0,1]
      s_ += dpp_get<0x141, 0xF>(s_);     // row_half_mirror: the other quad of the group of 8
      if (part == 0) YV[I] = s_;
    }
    __syncthreads();
    double a0[1] = {0.0};
    for (int i = tid; i < n; i += WG) {
      const double ri = r[i], z = ri + YV[AG[i]];
      q[i] = z;
      a0[0] += ri * z;
    }
    block_sum1<1>(a0, red, rsel);
    const double rz = a0[0];
    const double beta = rz / rz_old;
    rz_old = rz;
    if (it == 0)
      for (int i = tid; i < n; i += WG) p[i] = q[i];
    else
      for (int i = tid; i < n; i += WG) p[i] = q[i] + beta * p[i];
    ++it;
    __syncthreads();
    double a1[1] = {0.0};
    spmv_sell(sl_off, sl_col, A, p, n, [&](int row, double y0) {
      q[row] = y0;
      a1[0] += p[row] * y0;
    });
    block_sum1<1>(a1, red, rsel);
    if (!(a1[0] > 0.0)) break;
    const double alpha = rz / a1[0];
    double a2[1] = {0.0};
    for (int i = tid; i < n; i += WG) {
      x[i] += alpha * p[i];
      const double rn = r[i] - alpha * q[i];
      r[i] = rn;
      a2[0] += rn * rn;
    }
    block_sum1<1>(a2, red, rsel);
    rr = a2[0];
    if (!(rr > tol2)) break;
  }
  __syncthreads();
  return it;
}

// ------------------------------------------------------------------ two-level PCG with the MATRIX ON THE CHIP (round 6)
//
// cg_pressure_2l_lds keeps the four CG vectors in LDS and streams the matrix from L2 in every iteration: on the red-refined
// ys930 (3 322 rows, 25.5 k SELL entries = 300 KB) an iteration lasts 16.4 us, i.e. 18 GB/s - what ONE workgroup pulls out
// of L2 (HISTORY 8.5) - and the 171 iterations of a fresh mesh are 2.8 ms of the S3 step's flow leg.  Here the matrix does
// not move after the set-up: thread t owns rows t + k WG (k < K <= 7) = lane t % 64 of SELL slice 8 k + t / 64, so the slices
// of the first OC_KR row groups are held in REGISTERS by their owner (OC_WR entries per row: fp64 value + 16-bit column,
// padded with zero values) and the remaining slices are copied to LDS (fp64 values + 16-bit columns, the SELL layout as it
// is); x, r, q, z of the own rows live in registers, only the search direction p - the one vector that is gathered - in LDS,
// and the fp32 inverse of the coarse matrix in registers of the 448 threads of the coarse product.  An iteration is then
// LDS gathers + five workgroup barriers.  Same preconditioner (tl_build), same recurrences, same stopping test as
// cg_pressure_2l_lds; the row sums run over the SELL columns in the same order.  U: the kernel's LDS union; in: U[0..n) = x0,
// U[NVp..NVp+n) = b; out: U[0..n) = x.  Returns the iterations, or -1 WITHOUT having touched anything when the mesh does
// not fit (more than 7 row groups, a register slice wider than OC_WR, the LDS slices beyond U).
constexpr int OC_KR = 4, OC_WR = 10, OC_KMAX = 7;
__device__ __attribute__((noinline)) int cg_pressure_2l_onchip(int n, const int32_t* sl_off, const int32_t* sl_col, const double* A,
                                                                const double* coords, double rtol, int maxit, double* U, size_t U_bytes,
                                                                int NVp, double* red) {
  constexpr int NW = WG / 64;
  static_assert(NW == NAGX && NAG * 8 <= WG && OC_WR % 2 == 0, "one wave per strip; 8 threads per coarse row");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsl = (n + 63) >> 6, K = (nsl + NW - 1) / NW;
  if (n < 4 * NAG || K > OC_KMAX || n > 0xFFFF) return -1;
  // ---- LDS plan: p [NP] | WP [NW][NAG] | YV [NAG] | AG [n] | LDS slices: values, then columns
  const int NP = (n + 7) & ~7;
  const int s_l0 = min(OC_KR * NW, nsl);                       // first slice that lives in LDS
  const int e_l0 = sl_off[s_l0], ent_l = sl_off[nsl] - e_l0;   // its first entry; entries in LDS
  double* P = U;
  double* WP = P + NP;
  double* YV = WP + NW * NAG;
  unsigned char* AG = reinterpret_cast<unsigned char*>(YV + NAG);
  double* LV = reinterpret_cast<double*>(AG + NP);
  unsigned short* LC = reinterpret_cast<unsigned short*>(LV + ent_l);
  const size_t need = sizeof(double) * (size_t)(NP + NW * NAG + NAG + ent_l) + (size_t)NP + sizeof(unsigned short) * (size_t)(ent_l + 8);
  // (the set-up's scratch - histograms, the fp64 coarse matrix, its fp32 inverse - lies where the LDS slices go afterwards)
  const size_t setup = sizeof(double) * (size_t)(NP + NW * NAG + NAG) + (size_t)NP + sizeof(double) * NAG * NAG + sizeof(float) * NAG * NAG +
                       sizeof(int) * (TL_XB + NAGX * TL_YB + 2 * NAGX) + (size_t)NP;
  bool bad = need > U_bytes || setup > U_bytes;
#pragma unroll
  for (int k = 0; k < OC_KR; ++k) {
    const int s_ = k * NW + wave;
    if (s_ < nsl && ((sl_off[s_ + 1] - sl_off[s_]) >> 6) > OC_WR) bad = true;
  }
  if (__syncthreads_or(bad)) return -1;
  // ---- own rows of x0 and b
  double x[OC_KMAX], r[OC_KMAX], q[OC_KMAX];
#pragma unroll
  for (int k = 0; k < OC_KMAX; ++k) {
    const int row = tid + k * WG;
    x[k] = row < n ? U[row] : 0.0;
    r[k] = row < n ? U[NVp + row] : 0.0;
  }
  int lb[OC_KMAX - OC_KR], lw[OC_KMAX - OC_KR];                 // LDS slices of this wave: first entry (relative), width
#pragma unroll
  for (int k = OC_KR; k < OC_KMAX; ++k) {
    const int s_ = k * NW + wave;
    lb[k - OC_KR] = __builtin_amdgcn_readfirstlane(s_ < nsl ? sl_off[s_] - e_l0 : 0);
    lw[k - OC_KR] = __builtin_amdgcn_readfirstlane(s_ < nsl ? (sl_off[s_ + 1] - sl_off[s_]) >> 6 : 0);
  }
  __syncthreads();                                              // (b is in registers: its LDS is free from here on)
  // ---- the preconditioner (scratch where the LDS slices go), its fp32 inverse into registers
  float ai[(NAG + 7) / 8];
  {
    double* AC = LV;
    float* AI = reinterpret_cast<float*>(AC + NAG * NAG);
    int* hist = reinterpret_cast<int*>(AI + NAG * NAG);
    tl_build(n, sl_off, sl_col, A, coords, hist, AC, AI, AG, red);
    __syncthreads();
    const int I = tid >> 3, part = tid & 7;
#pragma unroll
    for (int m = 0; m < (NAG + 7) / 8; ++m) ai[m] = (tid < NAG * 8 && part + 8 * m < NAG) ? AI[I * NAG + part + 8 * m] : 0.0f;
  }
  unsigned agp[2] = {0u, 0u};                                   // aggregates of the own rows, one byte each
#pragma unroll
  for (int k = 0; k < OC_KMAX; ++k) agp[k >> 2] |= (tid + k * WG < n ? (unsigned)AG[tid + k * WG] : 0u) << (8 * (k & 3));
  auto ag = [&](int k) { return (int)((agp[k >> 2] >> (8 * (k & 3))) & 0xFFu); };
  __syncthreads();
  // ---- the LDS slices
  for (int e = tid; e < ent_l; e += WG) {
    LV[e] = A[e_l0 + e];
    LC[e] = (unsigned short)sl_col[e_l0 + e];
  }
  __syncthreads();
  // (the register slices are loaded LAST: across the set-up above they were the allocator's first spill candidates)
  double av[OC_KR][OC_WR];
  unsigned ac[OC_KR][OC_WR / 2];
#pragma unroll
  for (int k = 0; k < OC_KR; ++k) {
    const int s_ = k * NW + wave, own = min(tid + k * WG, n - 1);
    const int base = s_ < nsl ? sl_off[s_] : 0, wd = s_ < nsl ? (sl_off[s_ + 1] - base) >> 6 : 0;
#pragma unroll
    for (int j = 0; j < OC_WR; j += 2) {
      const bool h0 = j < wd, h1 = j + 1 < wd;
      av[k][j] = h0 ? A[base + lane + j * 64] : 0.0;
      av[k][j + 1] = h1 ? A[base + lane + (j + 1) * 64] : 0.0;
      const unsigned c0 = h0 ? (unsigned)sl_col[base + lane + j * 64] : (unsigned)own;
      const unsigned c1 = h1 ? (unsigned)sl_col[base + lane + (j + 1) * 64] : (unsigned)own;
      ac[k][j / 2] = c0 | (c1 << 16);
    }
  }
  auto spmv = [&](double (&y)[OC_KMAX]) {
#pragma unroll
    for (int k = 0; k < OC_KR; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int j = 0; j < OC_WR; j += 2) {
        // (opaque: the 40 gather addresses are loop invariants the compiler would otherwise keep in 40 registers - and spill)
        asm volatile("" : "+v"(ac[k][j / 2]));
        const unsigned c = ac[k][j / 2];
        s_ += av[k][j] * P[c & 0xFFFFu];
        s_ += av[k][j + 1] * P[c >> 16];
      }
      y[k] = s_;
      __builtin_amdgcn_sched_barrier(0);       // (row by row: all 40 gathers in flight at once cost 80 registers and spilled)
    }
#pragma unroll
    for (int k = OC_KR; k < OC_KMAX; ++k) {
      double s_ = 0.0;
      const double* a = LV + lb[k - OC_KR] + lane;
      const unsigned short* c = LC + lb[k - OC_KR] + lane;
      for (int j = 0; j < lw[k - OC_KR]; ++j) s_ += a[j * 64] * P[c[j * 64]];
      y[k] = s_;
    }
  };
  // ---- r = b - A x0
  double acc[2] = {0.0, 0.0};
  spmv(q);
#pragma unroll
  for (int k = 0; k < OC_KMAX; ++k)
    if (tid + k * WG < n) {
      const double b = r[k], r0 = b - q[k];
      r[k] = r0;
      acc[0] += b * b;
      acc[1] += r0 * r0;
    }
  block_sum<2>(acc, red);
  const double bb = acc[0], tol2 = rtol * rtol * bb;
  double rr = acc[1];
  int it = 0;
  if (rr > tol2 && bb != 0.0) {
    int rsel = 0;
    double rz_old = 1.0;
    while (it < maxit) {
      // z = r + P A_c^-1 P^T r
      if (lane < NAG) WP[wave * NAG + lane] = 0.0;
#pragma unroll
      for (int k = 0; k < OC_KMAX; ++k)
        if (tid + k * WG < n) atomicAdd(&WP[wave * NAG + ag(k)], r[k]);
      __syncthreads();
      if (tid < NAG * 8) {
        const int I = tid >> 3, part = tid & 7;
        int part_o = part;
        asm volatile("" : "+v"(part_o));            // (one base address + immediate offsets instead of 56 hoisted addresses)
        const double* wq = WP + part_o;
        double s_ = 0.0;
        static_assert(NAG % 8 == 0, "every thread of a coarse row has NAG / 8 columns");
#pragma unroll
        for (int m = 0; m < NAG / 8; ++m) {
          double w_ = 0.0;
#pragma unroll
          for (int w2 = 0; w2 < NW; ++w2) w_ += wq[w2 * NAG + 8 * m];
          s_ += (double)ai[m] * w_;
        }
        s_ += dpp_get<0xB1, 0xF>(s_);      // quad_perm [1,0,3,2]
        s_ += dpp_get<0x4E, 0xF>(s_);      // quad_perm [2,3,0,1]
        s_ += dpp_get<0x141, 0xF>(s_);     // row_half_mirror: the other quad of the group of 8
        if (part == 0) YV[I] = s_;
      }
      __syncthreads();
      double a0[1] = {0.0};
#pragma unroll
      for (int k = 0; k < OC_KMAX; ++k)
        if (tid + k * WG < n) a0[0] += r[k] * (r[k] + YV[ag(k)]);
      block_sum1<1>(a0, red, rsel);
      const double rz = a0[0];
      const double beta = rz / rz_old;
      rz_old = rz;
#pragma unroll
      for (int k = 0; k < OC_KMAX; ++k)
        if (tid + k * WG < n) {
          const double z = r[k] + YV[ag(k)];          // (read again instead of kept across the reduction: 14 registers)
          P[tid + k * WG] = it == 0 ? z : z + beta * P[tid + k * WG];
        }
      ++it;
      __syncthreads();
      double a1[1] = {0.0};
      spmv(q);
#pragma unroll
      for (int k = 0; k < OC_KMAX; ++k)
        if (tid + k * WG < n) a1[0] += P[tid + k * WG] * q[k];
      block_sum1<1>(a1, red, rsel);
      if (!(a1[0] > 0.0)) break;
      const double alpha = rz / a1[0];
      double a2[1] = {0.0};
#pragma unroll
      for (int k = 0; k < OC_KMAX; ++k)
        if (tid + k * WG < n) {
          x[k] += alpha * P[tid + k * WG];
          const double rn = r[k] - alpha * q[k];
          r[k] = rn;
          a2[0] += rn * rn;
        }
      block_sum1<1>(a2, red, rsel);
      rr = a2[0];
      if (!(rr > tol2)) break;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OC_KMAX; ++k)
    if (tid + k * WG < n) U[tid + k * WG] = x[k];
  __syncthreads();
  return it;
}

// the Krylov pressure solve of the kernels whose pressure vectors live in LDS (modes 0 / 4 / 5 / 7): two-level from
// TL_AUTO_NV vertices on (pcg_degree 0 = auto) or on request (pcg_degree < 0), Jacobi-CG otherwise (pcg_degree > 0: always)
#ifndef MDQ_TL_AUTO_NV
#define MDQ_TL_AUTO_NV 2048
#endif
__device__ inline int pressure_krylov_lds(const mdq_ipcs_desc& d, bool k1_lds, int n, const int32_t* sl_off, const int32_t* sl_col,
                                          const double* A, const double* coords, double* x, double* r, double* p, double* q,
                                          size_t scratch_bytes, double* extra, size_t extra_bytes, double* red,
                                          double* U = nullptr, size_t U_bytes = 0, int NVp = 0) {
  int it = -1;
  // (U: the LDS union of the kernel with x = U, r = U + NVp: the matrix-on-chip solver where the mesh fits it; pcg_degree -2
  //  keeps round 5's solver with the matrix streamed from L2 - A / B switch)
  if (!k1_lds && U != nullptr && d.pcg_degree != -2 && (d.pcg_degree < 0 || (d.pcg_degree == 0 && n >= MDQ_TL_AUTO_NV)))
    it = cg_pressure_2l_onchip(n, sl_off, sl_col, A, coords, d.rtol, d.maxit_p, U, U_bytes, NVp, red);
  if (it < 0 && !k1_lds && (d.pcg_degree < 0 || (d.pcg_degree == 0 && n >= MDQ_TL_AUTO_NV)))
    it = cg_pressure_2l_lds(n, sl_off, sl_col, A, coords, d.rtol, d.maxit_p, x, r, p, q, scratch_bytes, extra, extra_bytes, red);
  if (it < 0) it = cg_pressure(n, sl_off, sl_col, A, d.rtol, d.maxit_p, x, r, p, q, red);
  return it;
}

#ifdef MDQ_AT_TRACE
// debug build only: s_memtime deltas of thread 0 of environment 0 at the phase boundaries of at_velocity_kernel
static __device__ long long mdq_at_trace_buf[16];
#define AT_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (tid == 0 && b == 0) mdq_at_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
#if MDQ_IN_PART(0)
extern "C" MDQ_API int mdq_at_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_at_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_at_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
static __device__ long long mdq_pt_trace_buf[16];
#define PT_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) mdq_pt_trace_buf[k] += tn_ - tqp_; tqp_ = tn_; }
#if MDQ_IN_PART(0)
extern "C" MDQ_API int mdq_pt_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_pt_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_pt_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
static __device__ long long mdq_ct_trace_buf[16];
#define CT_STAMP(k) { const long long tn_ = __builtin_amdgcn_s_memtime(); if (tid == 0 && b == 0) mdq_ct_trace_buf[k] += tn_ - tq_; tq_ = tn_; }
#if MDQ_IN_PART(0)
extern "C" MDQ_API int mdq_ct_trace_host(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdq_ct_trace_buf), sizeof(long long) * 16) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mdq_ct_trace_buf), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif
#else
#define AT_STAMP(k)
#define CT_STAMP(k)
#define PT_STAMP(k)
#endif

// ================================================================== direct pressure solve
//
// Solve phase of the substructuring factorisation built by meshdqn_amd/pressure_direct.py
// (the MI355X counterpart of the MUMPS back-substitution of flow_solver.py:380): dense
// column-major blocks, one thread per row, coalesced streaming, three barriers per solve.
struct PdView {
  int nI, nG, nparts;
  const int32_t *node, *meta, *rowblk, *gidx, *gk_ptr, *gk_col;
  const double *W, *F, *Sinv, *gk_val;
};

__device__ __forceinline__ PdView pd_view(const mdq_ipcs_desc& d, int b) {
  PdView p;
  const int64_t B = b;
  const int32_t* h = d.pd_hdr + B * 4;
  p.nI = h[0];
  p.nG = h[1];
  p.nparts = h[2];
  p.node = d.pd_node + B * d.NV;
  p.meta = d.pd_meta + B * d.NPART * 6;
  p.rowblk = d.pd_rowblk + B * d.NV;
  p.W = d.pd_W + B * d.NPW;
  p.F = d.pd_F + B * d.NPF;
  p.gidx = d.pd_gidx + B * d.NPGI;
  p.Sinv = d.pd_Sinv + B * d.NPS;
  p.gk_ptr = d.pd_gk_ptr + B * (d.NV + 1);
  p.gk_col = d.pd_gk_col + B * d.NPGK;
  p.gk_val = d.pd_gk_val + B * d.NPGK;
  return p;
}

// sum_j A[j * stride] * x[idx(j)], j in [0, len): the strided global loads in batches of PDU, ALL of a batch in flight
// together; the last batch is padded with clamped (re-read, weight 0) entries, so there is no serial remainder loop
// (with `#pragma unroll 8/16` the remainder of a 55-column block is up to 15 dependent round trips to the memory side).
constexpr int PDU = 16;
template <class Idx>
__device__ __forceinline__ double pd_dot(const double* __restrict__ A, int64_t stride, int len, const double* x, Idx idx) {
  double acc = 0.0;
  for (int j0 = 0; j0 < len; j0 += PDU) {
    double av[PDU];
    int xi[PDU];
#pragma unroll
    for (int u = 0; u < PDU; ++u) {
      const int j = min(j0 + u, len - 1);
      av[u] = A[(int64_t)j * stride];
      xi[u] = idx(j);
    }
#pragma unroll
    for (int u = 0; u < PDU; ++u) acc += (j0 + u < len ? av[u] : 0.0) * x[xi[u]];
  }
  return acc;
}

// x = K^-1 b.  b, x: LDS vectors in natural node order (x may alias b); t0,t1,t2: LDS scratch
// of at least max(n, NTH) doubles each (NTH = threads of the calling kernel).
template <int NTH = WG>
__device__ inline void pressure_direct(const PdView& pd, int n, const double* b, double* x, double* t0, double* t1,
                                       double* t2) {
  const int tid = threadIdx.x, nI = pd.nI, nG = pd.nG;
  double* bp = t0;  // permuted right-hand side
  double* y = t1;   // W b_I  |  separator: g, then x_G
#ifdef MDQ_AT_TRACE
  long long tqp_ = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  for (int q = tid; q < n; q += NTH) bp[q] = b[pd.node[q]];
  __syncthreads();
  PT_STAMP(1)
  // y_I = W b_I
  for (int q = tid; q < nI; q += NTH) {
    const int32_t* m6 = pd.meta + 6 * pd.rowblk[q];
    const int q0 = m6[0], m = m6[1];
    const double* Wc = pd.W + m6[2] + (q - q0);
    y[q] = pd_dot(Wc, m, m, bp, [&](int j) { return q0 + j; });
  }
  __syncthreads();
  PT_STAMP(2)
  // g = b_G - K[G,I] y_I
  for (int g = tid; g < nG; g += NTH) {
    double acc = bp[nI + g];
    for (int k = pd.gk_ptr[g]; k < pd.gk_ptr[g + 1]; ++k) acc -= pd.gk_val[k] * y[pd.gk_col[k]];
    y[nI + g] = acc;
  }
  __syncthreads();
  PT_STAMP(3)
  // x_G = Sinv g : rows split over column slices so that all waves stream Sinv
  const int NGP = (nG + 63) & ~63;
  const int nsl = NGP > 0 ? (NTH / NGP > 0 ? NTH / NGP : 1) : 1;
  const int cw = (nG + nsl - 1) / nsl;
  for (int idx = tid; idx < nsl * NGP; idx += NTH) {
    const int sl = idx / NGP, row = idx - sl * NGP;
    double acc = 0.0;
    if (row < nG) {
      const int c1 = min(nG, (sl + 1) * cw);
      const double* Sc = pd.Sinv + row;
#pragma unroll 8
      for (int c = sl * cw; c < c1; ++c) acc += Sc[(int64_t)c * nG] * y[nI + c];   // (pd_dot was slower here: 13 columns)
    }
    t2[idx] = acc;
  }
  __syncthreads();
  for (int g = tid; g < nG; g += NTH) {
    double acc = 0.0;
    for (int sl = 0; sl < nsl; ++sl) acc += t2[sl * NGP + g];
    bp[g] = acc;  // x_G (bp is free now)
  }
  __syncthreads();
  PT_STAMP(4)
  // x_I = y_I - F x_G ; scatter back to natural order
  for (int q = tid; q < nI; q += NTH) {
    const int32_t* m6 = pd.meta + 6 * pd.rowblk[q];
    const int q0 = m6[0], m = m6[1], gs = m6[4];
    const double* Fc = pd.F + m6[3] + (q - q0);
    const int32_t* gi = pd.gidx + m6[5];
    double acc = y[q];
    if (gs > 0) acc -= pd_dot(Fc, m, gs, bp, [&](int c) { return gi[c]; });
    x[pd.node[q]] = acc;
  }
  for (int g = tid; g < nG; g += NTH) x[pd.node[nI + g]] = bp[g];
  __syncthreads();
  PT_STAMP(5)
}

// ================================================================== probes

// (drag, lift) of one field pair on the airfoil facets; result valid in every thread.
__device__ inline void forces(const EnvView& v, double mu, const double2* __restrict__ u,
                              const double* __restrict__ p, double* red, double& drag, double& lift) {
  double acc[2] = {0.0, 0.0};
  for (int f = threadIdx.x; f < v.naf; f += WG) {
    const int e = v.af_facets[2 * f], k = v.af_facets[2 * f + 1];
    const ElemIdx E = load_dofs(v, e);
    double X[3][2];
    load_cell_coords(v, e, X);
    // geometry straight from the coordinates: the probes then need no assembled operators
    // (Env2DAirfoil.calculate_reward samples them on every freshly coarsened mesh)
    Geo g;
    {
      const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0];
      const double J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
      const double det = J00 * J11 - J01 * J10;
      g.j00 = J11 / det;
      g.j01 = -J01 / det;
      g.j10 = -J10 / det;
      g.j11 = J00 / det;
      g.det = fabs(det);
    }
    const Facet F = facet_geometry(X, k);
    double2 ue[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ue[i] = u[E.dof[i]];
    double pe[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pe[i] = p[E.dof[i]];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const double s = (q == 0 ? 0.5 - 0.28867513459481288225 : 0.5 + 0.28867513459481288225), w = 0.5 * F.len;
      const double xi = F.ra[0] + s * (F.rb[0] - F.ra[0]);
      const double eta = F.ra[1] + s * (F.rb[1] - F.ra[1]);
      double phi[6], dphi[6][2];
      p2_eval(xi, eta, phi, dphi);
      double uxx = 0, uxy = 0, uyx = 0, uyy = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double gx = g.j00 * dphi[j][0] + g.j10 * dphi[j][1];
        const double gy = g.j01 * dphi[j][0] + g.j11 * dphi[j][1];
        uxx += ue[j].x * gx;
        uxy += ue[j].x * gy;
        uyx += ue[j].y * gx;
        uyy += ue[j].y * gy;
      }
      const double pq = pe[0] * (1.0 - xi - eta) + pe[1] * xi + pe[2] * eta;
      const double sxx = 2.0 * mu * uxx - pq, sxy = mu * (uxy + uyx), syy = 2.0 * mu * uyy - pq;
      acc[0] += w * (sxx * F.nx + sxy * F.ny);
      acc[1] += w * (sxy * F.nx + syy * F.ny);
    }
  }
  block_sum<2>(acc, red);
  drag = acc[0];
  lift = acc[1];
}

#if MDQ_IN_PART(0)
static __global__ __launch_bounds__(WG) void probe_kernel(mdq_ipcs_desc d, int nfields, const double* u, const double* p,
                                                    double* drag, double* lift) {
  __shared__ double red[NWAVE * 2];
  const int b = blockIdx.x;
  const EnvView v = env_view(d, b);
  for (int f = 0; f < nfields; ++f) {
    const double2* uf = reinterpret_cast<const double2*>(u) + ((int64_t)b * nfields + f) * d.N2;
    const double* pf = p + ((int64_t)b * nfields + f) * d.NV;
    double dr, li;
    forces(v, d.mu, uf, pf, red, dr, li);
    if (threadIdx.x == 0) {
      drag[(int64_t)b * nfields + f] = dr;
      lift[(int64_t)b * nfields + f] = li;
    }
    __syncthreads();
  }
}
#endif

// ================================================================== time stepping

#ifdef MDQ_PROFILE
#define MDQ_STAMP(k) { __syncthreads(); long long tn = __builtin_amdgcn_s_memtime(); prof[k] += tn - tprev; tprev = tn; }
#else
#define MDQ_STAMP(k)
#endif

// dynamic LDS: [red 64 doubles][union: velocity stage (MODE 1: p,r  double2[N2p] each;
//                                          MODE 2: x stage double2[N2p] + element tile double2[6*WG])
//                                  | pressure: 4 vectors [NVp] + K1 in SELL form (values, columns, slice offsets)]
struct LdsPlan {
  int N2p, NVp;
  size_t vel1_bytes, vel2_bytes, vel3_bytes, prs_vec_bytes, prs_mat_bytes;
};
__host__ __device__ inline LdsPlan lds_plan(int N2, int NV, int NSE1) {
  LdsPlan P;
  P.N2p = (N2 + 1) & ~1;
  P.NVp = ((NV > 1024 ? NV : 1024) + 63) & ~63;  // (>= the 1024 threads of the direct pressure kernel)
  P.vel1_bytes = 2 * sizeof(double2) * (size_t)P.N2p;
  P.vel2_bytes = sizeof(double2) * ((size_t)P.N2p + 6 * MF_CH);  // stage + tile
  P.vel3_bytes = 3 * sizeof(double2) * (size_t)P.N2p;  // p, r, result vector
  P.prs_vec_bytes = 5 * sizeof(double) * (size_t)P.NVp;  // x, r, p, q + one scratch vector (direct solver)
  P.prs_mat_bytes = sizeof(double) * (size_t)(NSE1 > WG ? NSE1 : WG) + sizeof(int32_t) * ((size_t)NSE1 + (NSE1 & 1)) +
                    sizeof(int32_t) * (size_t)(((NV / 64 + 2) + 1) & ~1);
  return P;
}

// PG: the four pressure CG vectors live in the workspace slab as well (they alias the velocity Krylov vectors, idle during
// step 2) - a mesh whose pressure vectors exceed the LDS (NV > ~4000) still steps; modes 0 / 5 only, K1_LDS = false.
template <int MODE, bool K1_LDS, bool PG = false>
__global__ __launch_bounds__(WG) void evolve_kernel(mdq_ipcs_desc d, int nsteps, double* drag, double* lift,
                                                     int32_t* iters) {
  static_assert(!PG || (!K1_LDS && (MODE == 0 || MODE == 5)), "global pressure vectors: modes 0 / 5 without the LDS matrix");
  extern __shared__ __align__(16) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);

  double* red = smem;  // 64 doubles
  double* U = smem + 64;

  // workspace carve-up (global; stays in this CU's L2 slice)
  double* w = v.work;
  double2* escr2 = reinterpret_cast<double2*>(w);  // 6*NT double2 (aliased as 3*NT doubles for step 2)
  double* escr1 = w;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);
  double2* vr = xs + d.N2;
  double2* vh = vr + d.N2;
  double2* vp = vh + d.N2;
  double2* vv = vp + d.N2;
  double2* vt = vv + d.N2;
  // pressure view of the union (PG: of the idle velocity vectors; 2 N2 doubles each >= NV)
  double* px = PG ? reinterpret_cast<double*>(vr) : U;
  double* pr = PG ? reinterpret_cast<double*>(vh) : px + P.NVp;
  double* pp = PG ? reinterpret_cast<double*>(vp) : pr + P.NVp;
  double* pq = PG ? reinterpret_cast<double*>(vv) : pp + P.NVp;
  double* lK = pq + P.NVp;                                            // [NSE1]   (K1_LDS only)
  int32_t* lci = reinterpret_cast<int32_t*>(lK + d.NSE1);            // [NSE1]
  int32_t* lso = lci + d.NSE1 + (d.NSE1 & 1);                         // [NV/64+2]
  double2* h1 = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));  // u* history of the velocity solve
  double2* h2 = h1 + d.N2;
  double2* h3 = h2 + d.N2;
  double2* h4 = h3 + d.N2;
  double2* h5 = h4 + d.N2;
  double* hcnt = reinterpret_cast<double*>(h5 + d.N2);
  double* pnew = reinterpret_cast<double*>(vt + d.N2);
  // velocity view of the union
  double2* L0 = reinterpret_cast<double2*>(U);
  double2* L1 = L0 + P.N2p;

  VelCtx vc;
  vc.x = xs;
  vc.rh = vh;
  vc.vv = vv;
  vc.t = vt;
  vc.a = d.rho / d.dt;
  vc.mu = d.mu;
  // MF: the element-tile operators (mode 5; the MODE == 2 branches of this kernel are the same formulas with an LDS stage)
  constexpr bool MF = MODE == 2 || MODE == 5;
  double2* ytmp = reinterpret_cast<double2*>(w + work_ytmp_offset(d.NV, d.NT, d.NE));   // mode 5: accumulation vector
  double2* stage = MODE == 5 ? vh : L0;   // where the mass solve stages S^-1 p for the gathers (mode 5: vh is free there)
  vc.es = MODE == 5 ? L0 : L1;
  vc.yt = ytmp;
  vc.packed = d.N2 <= 4096;
  if (MODE == 1) {
    vc.p = vc.gp = L0;
    vc.r = vc.gr = L1;
  } else if (MODE == 2) {
    vc.p = vp;
    vc.r = vr;
    vc.gp = vc.gr = L0;
  } else {                       // modes 0 / 5: the gathers read the Krylov vectors themselves (global memory)
    vc.p = vc.gp = vp;
    vc.r = vc.gr = vr;
  }
  MassCtx mc;
  mc.x = xs;
  mc.r = vr;
  mc.q = vv;
  mc.es = MODE == 5 ? L0 : L1;
  mc.yt = ytmp;
  mc.packed = d.N2 <= 4096;
  if (MF) {
    mc.p = vp;
    mc.gp = stage;
  } else {
    mc.p = mc.gp = (MODE == 1) ? L0 : vp;
  }

  const int nsl1 = (nv + 63) >> 6;
  const int32_t* so1 = K1_LDS ? lso : v.sl1_off;
  const int32_t* ci1 = K1_LDS ? lci : v.sl1_col;
  const double* K1 = K1_LDS ? lK : v.K1s;

  int it_u = 0, it_p = 0, it_m = 0;
#ifdef MDQ_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    // ---------------- step 1: tentative velocity  (flow_solver.py:106-112, solve 1 of :378-380)
    MDQ_STAMP(7)
    rhs1_elements(v, d, v.u_n, v.p_n, escr2);
    __syncthreads();
    MDQ_STAMP(0)
    double acc[2] = {0.0, 0.0};
    const int nhist = (int)hcnt[0];
    for (int i = tid; i < n2; i += WG) {
      double2 f = make_double2(0.0, 0.0);
      for (int s = v.g2_ptr[i]; s < v.g2_ptr[i + 1]; ++s) {
        const double2 c = escr2[v.g2_src[s]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift1[i], id = v.idiag1[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) * id.x, (f.y - l.y) * id.y);
      // initial guess: polynomial extrapolation in time of the previous tentative velocities h1 = u*_n .. h5 (the
      // correction solve re-uses xs, so the history is kept separately; see at_velocity_kernel), u_n while there is
      // no history; Dirichlet values hold
      double2 x0 = v.u_n[i];
      if (nhist >= 2) {
        const double2 us1 = h1[i], us2 = h2[i];
        x0 = make_double2(2.0 * us1.x - us2.x, 2.0 * us1.y - us2.y);
        if (nhist >= 3) {
          const double2 us3 = h3[i];
          x0 = make_double2(3.0 * (us1.x - us2.x) + us3.x, 3.0 * (us1.y - us2.y) + us3.y);
          if (nhist >= 4) {
            const double2 us4 = h4[i];
            x0 = make_double2(4.0 * (us1.x + us3.x) - 6.0 * us2.x - us4.x, 4.0 * (us1.y + us3.y) - 6.0 * us2.y - us4.y);
            if (nhist >= 5) {
              const double2 us5 = h5[i];
              x0 = make_double2(5.0 * (us1.x - us4.x) - 10.0 * (us2.x - us3.x) + us5.x,
                                5.0 * (us1.y - us4.y) - 10.0 * (us2.y - us3.y) + us5.y);
            }
          }
        }
      }
      if (fl) x0 = g;
      xs[i] = x0;
      if (MF) {
        if (MODE == 2) L0[i] = x0;
        vc.r[i] = fl ? make_double2(0.0, 0.0) : make_double2(f.x * id.x, f.y * id.y);
      } else {
        vc.r[i] = bi;
      }
      acc[0] += bi.x * bi.x + bi.y * bi.y;
    }
    __syncthreads();
    apply_velocity<MODE>(v, vc, MODE == 2 ? L0 : xs, [&](int row, double y0, double y1) {   // (mode 5 gathers x0 from xs)
      const double2 bi = vc.r[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      vc.r[row] = r0;
      vc.rh[row] = r0;
      vc.p[row] = make_double2(0.0, 0.0);
      vc.vv[row] = make_double2(0.0, 0.0);
      acc[1] += r0.x * r0.x + r0.y * r0.y;
    });
    block_sum<2>(acc, red);
    MDQ_STAMP(1)
    it_u += bicgstab_velocity<MODE>(v, vc, d.rtol, d.maxit_u, acc[0], acc[1], red);
    __syncthreads();
    for (int i = tid; i < n2; i += WG) {  // shift the history, newest first: h1 = u* of this step
      if (nhist >= 4) h5[i] = h4[i];
      if (nhist >= 3) h4[i] = h3[i];
      if (nhist >= 2) h3[i] = h2[i];
      if (nhist >= 1) h2[i] = h1[i];
      h1[i] = xs[i];
    }
    if (tid == 0) hcnt[0] = (double)(nhist < 5 ? nhist + 1 : 5);
    MDQ_STAMP(2)

    // ---------------- step 2: pressure  (flow_solver.py:115-116)
    if (K1_LDS && !d.pd_enabled) {
      const int ne1 = v.sl1_off[nsl1];
      for (int k = tid; k < ne1; k += WG) {
        lK[k] = v.K1s[k];
        lci[k] = v.sl1_col[k];
      }
      for (int k = tid; k <= nsl1; k += WG) lso[k] = v.sl1_off[k];
    }
    rhs2_elements(v, d, xs, v.p_n, escr1);
    __syncthreads();
    for (int i = tid; i < nv; i += WG) {
      double bsum = 0.0;
      for (int s = v.g1_ptr[i]; s < v.g1_ptr[i + 1]; ++s) bsum += escr1[v.g1_src[s]];
      const double sd = v.sdiagK[i];
      pr[i] = v.bcp_flag[i] ? 0.0 : bsum / sd;
      px[i] = v.p_n[i] * sd;
    }
    MDQ_STAMP(3)
    if (!PG && d.pd_enabled && d.pd_hdr[4 * (int64_t)b + 2] > 0) {   // (nparts = 0: no factors for this environment -> Krylov)
      const PdView pd = pd_view(d, b);
      pressure_direct(pd, nv, pr, px, pp, pq, lK);
    } else {
      if constexpr (PG)      // (vectors in the slab - pp is a 2 N2-double vector there -, the preconditioner's tables in the LDS they leave free)
        it_p += pressure_krylov_lds(d, false, nv, so1, ci1, K1, v.coords, px, pr, pp, pq, 2 * sizeof(double) * (size_t)d.N2, U,
                                    MODE == 5 ? sizeof(double2) * 6 * MF_CH : tl_extra_bytes(d.NV) + sizeof(double) * NAG * NAG, red);
      else
        it_p += pressure_krylov_lds(d, K1_LDS, nv, so1, ci1, K1, v.coords, px, pr, pp, pq, 2 * sizeof(double) * (size_t)P.NVp, lK,
                                    sizeof(double) * (size_t)P.NVp, red, MODE == 5 ? U : nullptr,     // (matrix on the chip: the tile modes only)
                                    max(P.prs_vec_bytes + (K1_LDS ? P.prs_mat_bytes : 0), MODE == 5 ? tile_lds_bytes(d) : (size_t)0), P.NVp);
    }
    MDQ_STAMP(4)
    for (int i = tid; i < nv; i += WG) pnew[i] = px[i] / v.sdiagK[i];
    __syncthreads();

    // ---------------- step 3: velocity correction  (flow_solver.py:119-120)
    rhs3_elements(v, d, xs, pnew, v.p_n, escr2);
    __syncthreads();
    double am[2] = {0.0, 0.0};
    for (int i = tid; i < n2; i += WG) {
      double2 f = make_double2(0.0, 0.0);
      for (int s = v.g2_ptr[i]; s < v.g2_ptr[i + 1]; ++s) {
        const double2 c = escr2[v.g2_src[s]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift3[i];
      const double sd = v.sdiagM[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) / sd, (f.y - l.y) / sd);
      const double2 x0 = fl ? g : xs[i];
      xs[i] = make_double2(x0.x * sd, x0.y * sd);  // scaled unknown S x
      if (MF) {
        stage[i] = x0;  // S^-1 (S x0)
        mc.r[i] = fl ? make_double2(0.0, 0.0) : make_double2(f.x / sd, f.y / sd);
      } else {
        mc.r[i] = bi;
      }
      am[0] += bi.x * bi.x + bi.y * bi.y;
    }
    __syncthreads();
    apply_mass<MODE>(v, mc, MF ? stage : xs, [&](int row, double y0, double y1) {
      const double2 bi = mc.r[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      mc.r[row] = r0;
      mc.p[row] = r0;
      am[1] += r0.x * r0.x + r0.y * r0.y;
    });
    block_sum<2>(am, red);
    if (MF) {
      // stage S^-1 p0 (own rows; the apply above is complete: block_sum barriers passed)
      for (int i = tid; i < n2; i += WG) {
        const double is = 1.0 / v.sdiagM[i];
        const double2 p0 = mc.p[i];
        stage[i] = make_double2(p0.x * is, p0.y * is);
      }
      __syncthreads();
    }
    MDQ_STAMP(5)
    it_m += cg_mass<MODE>(v, mc, d.rtol, d.maxit_m, am[0], am[1], red);
    MDQ_STAMP(6)

    // ---------------- update state + probes  (flow_solver.py:382-389)
    for (int i = tid; i < n2; i += WG) {
      const double sd = v.sdiagM[i];
      const double2 x = xs[i];
      v.u_n[i] = make_double2(x.x / sd, x.y / sd);
    }
    for (int i = tid; i < nv; i += WG) v.p_n[i] = pnew[i];
    __syncthreads();
    double dr, li;
    forces(v, d.mu, v.u_n, v.p_n, red, dr, li);
    if (tid == 0) {
      drag[(int64_t)b * nsteps + step] = dr;
      lift[(int64_t)b * nsteps + step] = li;
    }
  }
#ifdef MDQ_PROFILE
  if (tid == 0) {
    double* pw = pnew + d.NV;
    for (int k = 0; k < 8; ++k) pw[k] += (double)prof[k];
  }
#endif
  if (tid == 0 && iters) {
    iters[3 * b + 0] += it_u;
    iters[3 * b + 1] += it_p;
    iters[3 * b + 2] += it_m;
  }
}

template <int MODE, bool K1_LDS, bool PG = false>
static hipError_t launch_evolve(const mdq_ipcs_desc* d, size_t lds, int nsteps, double* drag, double* lift,
                                int32_t* iters, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_kernel<MODE, K1_LDS, PG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((evolve_kernel<MODE, K1_LDS, PG>), dim3(d->B), dim3(WG), lds, stream, *d, nsteps, drag, lift,
                     iters);
  return hipGetLastError();
}

// ================================================================== assembled operators, TWO workgroups per environment
//
// evolve_kernel<0> gives one workgroup - one CU - to an environment: with the BASELINE batch of 128 environments half of
// the chip idles, and on meshes whose operators no longer fit a CU's caches (BASELINE configs[4]: ys930 red-refined,
// 6 MB of velocity blocks per environment) the step is bound by what ONE CU pulls from L2 / HBM (~22 GB/s).  Here a TEAM
// of two workgroups shares an environment: element loops, row loops and SELL slices are dealt out alternately
// (global thread id = rank * WG + tid, stride 2 WG), all vectors live in the environment's workspace slab, and every
// barrier that separates a producer phase from a consumer phase is a TEAM barrier: agent-scope release by every wave,
// workgroup barrier, one arrival counter per team polled by one lane (bounded spin), workgroup barrier, agent-scope
// acquire - correct for any placement of the two workgroups (cdna_hip_programming.md, Guideline 16).  Reductions: each
// workgroup's deterministic block sum goes to the team's slot array and both workgroups add the two partials in rank
// order: every thread of the team sees the same bits.  The pressure solve (3 322 unknowns, LDS-resident vectors, direct
// factors or CG) stays with rank 0.  Same arithmetic as evolve_kernel<0> except for the association of the reductions.
constexpr int TEAM = 2;
struct Team {
  int rank;
  unsigned* ctr;     // [0] arrivals of the team, [1] set when a bounded spin ran out (the step's forces become NaN)
  double* slot;      // [2 parities][TEAM][4] partial sums
  unsigned epoch;
  int par;
  bool local;        // both workgroups run on the same XCD (team_place): the barrier need not go beyond that XCD's L2
  int rbeg, rend;    // mode 7: the rows of this workgroup in the vector passes and the epilogues
};
// The agent-scope fences of the general barrier write the XCD's L2 back and invalidate it - what an environment re-reads in
// every operator application (0.7 MB of element metadata on the refined mesh) is gone after every barrier: measured on the
// element tiles, 215 barriers per step, 27.3 ms per step of 128 refined meshes against 11.5 ms for ONE workgroup per
// environment.  Two workgroups on the SAME XCD share its L2, and the vector L1 of a CU is write-through: waiting for the
// own stores (vmcnt) and invalidating the own L1 behind the barrier is all the coherence they need.  The launcher's block
// numbering puts a team on block ids 8 apart - the same XCD under the dispatcher's round-robin - and every team CHECKS it
// (hardware register XCC_ID, exchanged through the general barrier at kernel start): any other placement keeps the
// general protocol.
__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xFu;
}
__device__ __forceinline__ void team_sync(Team& t) {
  if (t.local)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every wave: its stores have reached the XCD's L2
  else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // every wave: its stores are visible device-wide
  __syncthreads();
  ++t.epoch;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(t.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = t.epoch * TEAM;
    if (!__hip_atomic_load(t.ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      unsigned spins = 0;
      while (__hip_atomic_load(t.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22)) {                        // the partner never arrived (not resident?): give up, loudly
          __hip_atomic_store(t.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __syncthreads();
  if (t.local)
    asm volatile("buffer_inv sc0" ::: "memory");            // every wave: no stale lines of the partner's data in this CU's L1
  else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// the placement check at kernel start (through the general barrier); `ids`: two words of the environment's slab
__device__ __forceinline__ void team_place(Team& t, unsigned* ids, int general) {
  t.local = false;
  if (threadIdx.x == 0) __hip_atomic_store(ids + t.rank, xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  team_sync(t);
  const unsigned i0 = __hip_atomic_load(ids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned i1 = __hip_atomic_load(ids + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  static_assert(TEAM == 2, "two ids");
  t.local = i0 == i1 && !(general & 1);     // (`general` bit 0: MDQ_TEAM_GENERAL_BARRIER=1 keeps the placement-independent protocol - tests)
}
template <int N>
__device__ __forceinline__ void team_sum(double (&v)[N], double* red, Team& t) {
  static_assert(N <= 4, "slot capacity");
  block_sum<N>(v, red);
  double* mine = t.slot + (t.par * TEAM + t.rank) * 4;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int n = 0; n < N; ++n) mine[n] = v[n];
  }
  team_sync(t);
  const double* s0 = t.slot + (t.par * TEAM) * 4;
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = s0[n] + s0[4 + n];
  t.par ^= 1;
}

// A bounded spin of this team ran out in this launch (the partner workgroup was not resident for ~2^22 polls: another process
// or another stream held its CU).  Read by every thread behind a team barrier.  The step that sees it does NOT advance u_n / p_n,
// drops the initial-guess history (the vectors computed past the broken barrier are finite garbage), reports NaN forces for
// this and every later step of the launch, and the host raises (ipcs_batch.py / vec_env.py: `MeshDQNHipError`).
__device__ __forceinline__ bool team_failed(const Team& t) {
  return __hip_atomic_load(t.ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// bit 1 of the kernels' `general` argument (MDQ_TEAM_TEST_ABSENT_PARTNER=1, tests only): the second workgroup of
// environment 0 leaves at once - the time-out path of the barrier, forced
__device__ __forceinline__ bool team_test_absent(int general, int b, int rank) { return (general & 2) && b == 0 && rank == 1; }

// "No initial-guess history on a new mesh" (FlowSolver.remesh restarts u_n / p_n and the solvers, flow_solver.py:233-359):
// the counters of the extrapolated initial guesses (tentative velocities stored, corrections stored / ring position / lagged
// |b| / step parity of the fused correction start), of every operator mode, back to zero - instead of a fill of the whole
// workspace (100 MB per 128 environments in every S3 env step) - and, optionally, the iteration counters.
#if MDQ_IN_PART(0)
static __global__ void reset_history_kernel(mdq_ipcs_desc d, int32_t* iters) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  double* w = d.work + (int64_t)b * work_per_env(d.NV, d.NT, d.NE);
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);
  reinterpret_cast<double*>(xs + 3 * (int64_t)d.N2)[0] = 0.0;                       // mode 3: histc
  double2* h1 = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));
  double* ccnt = reinterpret_cast<double*>(h1 + 3 * (int64_t)d.N2);                  // mode 3: correction ring
  ccnt[0] = ccnt[1] = ccnt[2] = ccnt[3] = 0.0;
  reinterpret_cast<double*>(h1 + 5 * (int64_t)d.N2)[0] = 0.0;                        // modes 0-2, 4, 5: hcnt
  reinterpret_cast<double*>(h1 + 5 * (int64_t)d.N2)[1] = 0.0;                        // mode 2: corrections stored
  if (iters) iters[3 * b] = iters[3 * b + 1] = iters[3 * b + 2] = 0;
}
#endif

static __global__ void team_reset_kernel(mdq_ipcs_desc d);   // (defined in part 2 only; named by launch_evolve_team)
#if MDQ_IN_PART(2)
static __global__ void team_reset_kernel(mdq_ipcs_desc d) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  double* w = d.work + (int64_t)b * work_per_env(d.NV, d.NT, d.NE);
  double* spare = w + 12 * (int64_t)d.NT + 12 * (int64_t)d.N2 + d.NV;
  unsigned* ctr = reinterpret_cast<unsigned*>(spare + 8);
  ctr[0] = ctr[1] = 0u;
}
#endif

template <bool K1_LDS>
__global__ __launch_bounds__(WG) void evolve_team_kernel(mdq_ipcs_desc d, int nsteps, double* drag, double* lift,
                                                          int32_t* iters, int general) {
  extern __shared__ __align__(16) double smem[];
  // teams on neighbouring-by-8 block ids: with the dispatcher's round-robin both workgroups tend to share an XCD's L2
  // (a speed choice only: the protocol is placement-independent)
  const int q = blockIdx.x / (8 * TEAM), r8 = blockIdx.x % (8 * TEAM);
  const int b = q * 8 + (r8 & 7), rank = r8 >> 3;
  if (b >= d.B || team_test_absent(general, b, rank)) return;
  const int tid = threadIdx.x, gt = rank * WG + tid;
  constexpr int GS = TEAM * WG;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);
  double* red = smem;
  double* U = smem + 64;
  double* px = U;
  double* pr = px + P.NVp;
  double* pp = pr + P.NVp;
  double* pq = pp + P.NVp;
  double* lK = pq + P.NVp;
  int32_t* lci = reinterpret_cast<int32_t*>(lK + d.NSE1);
  int32_t* lso = lci + d.NSE1 + (d.NSE1 & 1);
  double* w = v.work;
  double2* escr2 = reinterpret_cast<double2*>(w);
  double* escr1 = w;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);
  double2* vr = xs + d.N2;
  double2* vh = vr + d.N2;
  double2* vp = vh + d.N2;
  double2* vv = vp + d.N2;
  double2* vt = vv + d.N2;
  double2* h1 = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));
  double2* h2 = h1 + d.N2;
  double2* h3 = h2 + d.N2;
  double2* h4 = h3 + d.N2;
  double2* h5 = h4 + d.N2;
  double* hcnt = reinterpret_cast<double*>(h5 + d.N2);
  double* pnew = reinterpret_cast<double*>(vt + d.N2);
  double* spare = pnew + d.NV;
  Team T;
  T.rank = rank;
  T.ctr = reinterpret_cast<unsigned*>(spare + 8);
  T.slot = spare + 9;
  T.epoch = 0;
  T.par = 0;
  const int nsl1 = (nv + 63) >> 6;
  const int32_t* so1 = K1_LDS ? lso : v.sl1_off;
  const int32_t* ci1 = K1_LDS ? lci : v.sl1_col;
  const double* K1 = K1_LDS ? lK : v.K1s;
  const double a = d.rho / d.dt;
  int it_u = 0, it_p = 0, it_m = 0;
  const int lane = tid & 63, gw = rank * NWAVE + (tid >> 6);
  constexpr int GW = TEAM * NWAVE;
  // SELL operator applications over the team's slices
  auto spmv_vel = [&](const double2* x, auto epi) {
    const int nsl = (n2 + 63) >> 6;
    const double4* A4 = reinterpret_cast<const double4*>(v.A1);
    for (int s = gw; s < nsl; s += GW) {
      const int base = v.sl2_off[s], wd = (v.sl2_off[s + 1] - base) >> 6;
      const double4* aa = A4 + base + lane;
      const int32_t* c = v.sl2_col + base + lane;
      double y0 = 0.0, y1 = 0.0;
#pragma unroll 4
      for (int j = 0; j < wd; ++j) {
        const double4 av = aa[j * 64];
        const double2 xv = x[c[j * 64]];
        y0 += av.x * xv.x + av.y * xv.y;
        y1 += av.z * xv.x + av.w * xv.y;
      }
      const int row = (s << 6) + lane;
      if (row < n2) epi(row, y0, y1);
    }
  };
  auto spmv_mass = [&](const double2* x, auto epi) {
    const int nsl = (n2 + 63) >> 6;
    for (int s = gw; s < nsl; s += GW) {
      const int base = v.sl2_off[s], wd = (v.sl2_off[s + 1] - base) >> 6;
      const double* aa = v.Ms + base + lane;
      const int32_t* c = v.sl2_col + base + lane;
      double y0 = 0.0, y1 = 0.0;
#pragma unroll 4
      for (int j = 0; j < wd; ++j) {
        const double av = aa[j * 64];
        const double2 xv = x[c[j * 64]];
        y0 += av * xv.x;
        y1 += av * xv.y;
      }
      const int row = (s << 6) + lane;
      if (row < n2) epi(row, y0, y1);
    }
  };
  team_place(T, reinterpret_cast<unsigned*>(spare + 25), general);
  for (int step = 0; step < nsteps; ++step) {
    // ---------------- step 1: tentative velocity
    for (int e = gt; e < v.nt; e += GS) {
      const ElemIdx E = load_dofs(v, e);
      const Geo g = load_geo(v, e);
      double2 ue[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ue[i] = v.u_n[E.dof[i]];
      double pe[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) pe[i] = v.p_n[E.dof[i]];
      double2 rr_[6];
      elem_rhs1_vol(g, a, d.mu, d.rho, ue, pe, rr_);
      const int ko = v.cell_outflow[e];
      if (ko >= 0) {
        double X[3][2];
        load_cell_coords(v, e, X);
        elem_outflow_add(g, X, ko, 0.5 * d.mu, ue, rr_);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) escr2[e * 6 + i] = rr_[i];
    }
    team_sync(T);
    double acc[2] = {0.0, 0.0};
    const int nhist = (int)hcnt[0];
    for (int i = gt; i < n2; i += GS) {
      double2 f = make_double2(0.0, 0.0);
      for (int s_ = v.g2_ptr[i]; s_ < v.g2_ptr[i + 1]; ++s_) {
        const double2 c = escr2[v.g2_src[s_]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift1[i], id = v.idiag1[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) * id.x, (f.y - l.y) * id.y);
      double2 x0 = v.u_n[i];
      if (nhist >= 2) {
        const double2 us1 = h1[i], us2 = h2[i];
        x0 = make_double2(2.0 * us1.x - us2.x, 2.0 * us1.y - us2.y);
        if (nhist >= 3) {
          const double2 us3 = h3[i];
          x0 = make_double2(3.0 * (us1.x - us2.x) + us3.x, 3.0 * (us1.y - us2.y) + us3.y);
          if (nhist >= 4) {
            const double2 us4 = h4[i];
            x0 = make_double2(4.0 * (us1.x + us3.x) - 6.0 * us2.x - us4.x, 4.0 * (us1.y + us3.y) - 6.0 * us2.y - us4.y);
            if (nhist >= 5) {
              const double2 us5 = h5[i];
              x0 = make_double2(5.0 * (us1.x - us4.x) - 10.0 * (us2.x - us3.x) + us5.x,
                                5.0 * (us1.y - us4.y) - 10.0 * (us2.y - us3.y) + us5.y);
            }
          }
        }
      }
      if (fl) x0 = g;
      xs[i] = x0;
      vr[i] = bi;
      acc[0] += bi.x * bi.x + bi.y * bi.y;
    }
    team_sync(T);
    spmv_vel(xs, [&](int row, double y0, double y1) {
      const double2 bi = vr[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      vr[row] = r0;
      vh[row] = r0;
      vp[row] = make_double2(0.0, 0.0);
      vv[row] = make_double2(0.0, 0.0);
      acc[1] += r0.x * r0.x + r0.y * r0.y;
    });
    team_sum<2>(acc, red, T);
    {  // BiCGStab (bicgstab_velocity<0>, rows and slices over the team)
      const double bb = acc[0], tol2 = d.rtol * d.rtol * bb;
      double rr = acc[1];
      if (rr > tol2 && bb != 0.0) {
        double rho = rr, rho_old = 1.0, alpha = 1.0, omega = 1.0;
        int it = 0;
        while (it < d.maxit_u) {
          ++it;
          const double beta = (rho / rho_old) * (alpha / omega);
          for (int i = gt; i < n2; i += GS) {
            const double2 ri = vr[i], pi = vp[i], vi = vv[i];
            vp[i] = make_double2(ri.x + beta * (pi.x - omega * vi.x), ri.y + beta * (pi.y - omega * vi.y));
          }
          team_sync(T);
          double a1[1] = {0.0};
          spmv_vel(vp, [&](int row, double y0, double y1) {
            vv[row] = make_double2(y0, y1);
            const double2 h = vh[row];
            a1[0] += h.x * y0 + h.y * y1;
          });
          team_sum<1>(a1, red, T);
          if (a1[0] == 0.0) break;
          alpha = rho / a1[0];
          double a2[1] = {0.0};
          for (int i = gt; i < n2; i += GS) {
            const double2 ri = vr[i], vi = vv[i];
            const double2 sv = make_double2(ri.x - alpha * vi.x, ri.y - alpha * vi.y);
            vr[i] = sv;
            a2[0] += sv.x * sv.x + sv.y * sv.y;
          }
          team_sum<1>(a2, red, T);       // (its team barrier also publishes s)
          if (!(a2[0] > tol2)) {
            for (int i = gt; i < n2; i += GS) {
              const double2 xi = xs[i], pi = vp[i];
              xs[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
            }
            break;
          }
          double a3[2] = {0.0, 0.0};
          spmv_vel(vr, [&](int row, double y0, double y1) {
            vt[row] = make_double2(y0, y1);
            const double2 sv = vr[row];
            a3[0] += y0 * sv.x + y1 * sv.y;
            a3[1] += y0 * y0 + y1 * y1;
          });
          team_sum<2>(a3, red, T);
          if (a3[1] == 0.0) break;
          omega = a3[0] / a3[1];
          double a4[2] = {0.0, 0.0};
          for (int i = gt; i < n2; i += GS) {
            const double2 ti = vt[i], xi = xs[i], pi = vp[i], si = vr[i], hi = vh[i];
            xs[i] = make_double2(xi.x + alpha * pi.x + omega * si.x, xi.y + alpha * pi.y + omega * si.y);
            const double2 rn = make_double2(si.x - omega * ti.x, si.y - omega * ti.y);
            vr[i] = rn;
            a4[0] += rn.x * rn.x + rn.y * rn.y;
            a4[1] += hi.x * rn.x + hi.y * rn.y;
          }
          team_sum<2>(a4, red, T);
          rr = a4[0];
          if (!(rr > tol2)) break;
          rho_old = rho;
          rho = a4[1];
          if (rho == 0.0 || omega == 0.0) break;
        }
        it_u += it;
      }
    }
    team_sync(T);
    for (int i = gt; i < n2; i += GS) {  // history, newest first
      if (nhist >= 4) h5[i] = h4[i];
      if (nhist >= 3) h4[i] = h3[i];
      if (nhist >= 2) h3[i] = h2[i];
      if (nhist >= 1) h2[i] = h1[i];
      h1[i] = xs[i];
    }
    // ---------------- step 2: pressure (element loop by the team, the solve by rank 0)
    {
      const double idt = 1.0 / d.dt;
      for (int e = gt; e < v.nt; e += GS) {
        const ElemIdx E = load_dofs(v, e);
        const Geo g = load_geo(v, e);
        double2 ue[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ue[i] = xs[E.dof[i]];
        double pe[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pe[i] = v.p_n[E.dof[i]];
        double r3[3];
        elem_rhs2(g, idt, ue, pe, r3);
#pragma unroll
        for (int j = 0; j < 3; ++j) escr1[e * 3 + j] = r3[j];
      }
    }
    team_sync(T);
    if (rank == 0) {
      if (tid == 0) hcnt[0] = (double)(nhist < 5 ? nhist + 1 : 5);
      if (K1_LDS && !d.pd_enabled) {
        const int ne1 = v.sl1_off[nsl1];
        for (int k = tid; k < ne1; k += WG) {
          lK[k] = v.K1s[k];
          lci[k] = v.sl1_col[k];
        }
        for (int k = tid; k <= nsl1; k += WG) lso[k] = v.sl1_off[k];
      }
      for (int i = tid; i < nv; i += WG) {
        double bsum = 0.0;
        for (int s_ = v.g1_ptr[i]; s_ < v.g1_ptr[i + 1]; ++s_) bsum += escr1[v.g1_src[s_]];
        const double sd = v.sdiagK[i];
        pr[i] = v.bcp_flag[i] ? 0.0 : bsum / sd;
        px[i] = v.p_n[i] * sd;
      }
      if (d.pd_enabled && d.pd_hdr[4 * (int64_t)b + 2] > 0) {
        const PdView pd = pd_view(d, b);
        pressure_direct(pd, nv, pr, px, pp, pq, lK);
      } else {
        it_p += pressure_krylov_lds(d, K1_LDS, nv, so1, ci1, K1, v.coords, px, pr, pp, pq, 2 * sizeof(double) * (size_t)P.NVp, lK,
                                    sizeof(double) * (size_t)P.NVp, red);
      }
      for (int i = tid; i < nv; i += WG) pnew[i] = px[i] / v.sdiagK[i];
    }
    team_sync(T);
    // ---------------- step 3: velocity correction
    for (int e = gt; e < v.nt; e += GS) {
      const ElemIdx E = load_dofs(v, e);
      const Geo g = load_geo(v, e);
      double2 ue[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ue[i] = xs[E.dof[i]];
      double dp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) dp[i] = pnew[E.dof[i]] - v.p_n[E.dof[i]];
      double2 rr_[6];
      elem_rhs3(g, d.dt, ue, dp, rr_);
#pragma unroll
      for (int i = 0; i < 6; ++i) escr2[e * 6 + i] = rr_[i];
    }
    team_sync(T);
    double am[2] = {0.0, 0.0};
    for (int i = gt; i < n2; i += GS) {
      double2 f = make_double2(0.0, 0.0);
      for (int s_ = v.g2_ptr[i]; s_ < v.g2_ptr[i + 1]; ++s_) {
        const double2 c = escr2[v.g2_src[s_]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift3[i];
      const double sd = v.sdiagM[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) / sd, (f.y - l.y) / sd);
      const double2 x0 = fl ? g : xs[i];
      xs[i] = make_double2(x0.x * sd, x0.y * sd);
      vr[i] = bi;
      am[0] += bi.x * bi.x + bi.y * bi.y;
    }
    team_sync(T);
    spmv_mass(xs, [&](int row, double y0, double y1) {
      const double2 bi = vr[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      vr[row] = r0;
      vp[row] = r0;
      am[1] += r0.x * r0.x + r0.y * r0.y;
    });
    team_sum<2>(am, red, T);
    {  // CG on the mass system (cg_mass<0>)
      const double bb = am[0], tol2 = d.rtol * d.rtol * bb;
      double rr = am[1];
      if (rr > tol2 && bb != 0.0) {
        int it = 0;
        while (it < d.maxit_m) {
          ++it;
          double a1[1] = {0.0};
          spmv_mass(vp, [&](int row, double y0, double y1) {
            vv[row] = make_double2(y0, y1);
            const double2 pi = vp[row];
            a1[0] += pi.x * y0 + pi.y * y1;
          });
          team_sum<1>(a1, red, T);
          if (!(a1[0] > 0.0)) break;
          const double alpha = rr / a1[0];
          double a2[1] = {0.0};
          for (int i = gt; i < n2; i += GS) {
            const double2 xi = xs[i], pi = vp[i], ri = vr[i], qi = vv[i];
            xs[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
            const double2 rn = make_double2(ri.x - alpha * qi.x, ri.y - alpha * qi.y);
            vr[i] = rn;
            a2[0] += rn.x * rn.x + rn.y * rn.y;
          }
          team_sum<1>(a2, red, T);
          const double rr_new = a2[0];
          if (!(rr_new > tol2)) break;
          const double beta = rr_new / rr;
          rr = rr_new;
          for (int i = gt; i < n2; i += GS) {
            const double2 ri = vr[i], pi = vp[i];
            vp[i] = make_double2(ri.x + beta * pi.x, ri.y + beta * pi.y);
          }
          team_sync(T);
        }
        it_m += it;
      }
    }
    team_sync(T);
    // ---------------- update state + probes
    if (team_failed(T)) {       // (see team_failed: the state stays the last good one, NaN forces from here on)
      if (rank == 0 && tid == 0) {
        hcnt[0] = 0.0;
        if (d.status) atomicOr(d.status + b, MDQ_IPCS_TEAM_TIMEOUT);
        for (int s_ = step; s_ < nsteps; ++s_) drag[(int64_t)b * nsteps + s_] = lift[(int64_t)b * nsteps + s_] = __builtin_nan("");
      }
      break;
    }
    for (int i = gt; i < n2; i += GS) {
      const double sd = v.sdiagM[i];
      const double2 x = xs[i];
      v.u_n[i] = make_double2(x.x / sd, x.y / sd);
    }
    for (int i = gt; i < nv; i += GS) v.p_n[i] = pnew[i];
    team_sync(T);
    if (rank == 0) {
      double dr, li;
      forces(v, d.mu, v.u_n, v.p_n, red, dr, li);
      if (tid == 0) {
        const bool failed = team_failed(T);
        drag[(int64_t)b * nsteps + step] = failed ? __builtin_nan("") : dr;
        lift[(int64_t)b * nsteps + step] = failed ? __builtin_nan("") : li;
      }
    }
  }
  if (rank == 0 && tid == 0 && iters) {
    iters[3 * b + 0] += it_u;
    iters[3 * b + 1] += it_p;
    iters[3 * b + 2] += it_m;
  }
}

// MDQ_TEAM_GENERAL_BARRIER=1 (read per launch): the teams keep the placement-independent agent-scope barrier even when both
// workgroups share an XCD - the same arithmetic, so the same bits (tests/test_ipcs_gpu.py), at 2-3x the step time
static int team_general_barrier() {
  return (std::getenv("MDQ_TEAM_GENERAL_BARRIER") != nullptr ? 1 : 0) | (std::getenv("MDQ_TEAM_TEST_ABSENT_PARTNER") != nullptr ? 2 : 0);
}

// CUs the team modes may count on when they are chosen AUTOMATICALLY: every workgroup of a team must be resident at the same
// time (the barrier spins), so the launch stream has to own that many CUs - the CUs of the stream's mask when it carries one
// (hipExtStreamCreateWithCUMask: meshdqn_amd/streams.py, MDQ_CU_PARTITION), and none at all when the process has been told
// that it shares the GPU with other processes (MDQ_SHARE_GPU: several ranks on one device).  An explicit mode 4 / 7 is the
// caller's decision; a time-out is reported either way (team_failed).
static int team_cus(hipStream_t st) {
  static const int ncu = [] {
    int dev_ = 0, n_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || hipDeviceGetAttribute(&n_, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess)
      return 0;
    return n_;
  }();
  if (std::getenv("MDQ_SHARE_GPU") != nullptr) return 0;
  uint32_t mask[16] = {0};
  if (hipExtStreamGetCUMask(st, 16, mask) != hipSuccess) {
    (void)hipGetLastError();
    return ncu;
  }
  int n = 0;
  for (int i = 0; i < 16; ++i) n += __builtin_popcount(mask[i]);
  return n > 0 && n < ncu ? n : ncu;
}

template <bool K1_LDS>
static hipError_t launch_evolve_team(const mdq_ipcs_desc* d, size_t lds, int nsteps, double* drag, double* lift,
                                     int32_t* iters, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_team_kernel<K1_LDS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(team_reset_kernel, dim3((d->B + 63) / 64), dim3(64), 0, stream, *d);
  const int teams8 = (d->B + 7) / 8;                       // blocks: groups of 8 environments x TEAM ranks
  hipLaunchKernelGGL((evolve_team_kernel<K1_LDS>), dim3(teams8 * 8 * TEAM), dim3(WG), lds, stream, *d, nsteps, drag, lift,
                     iters, team_general_barrier());
  return hipGetLastError();
}

// ================================================================== element tiles, TWO workgroups per environment (mode 7)
//
// Mode 5 gives an environment ONE workgroup, and what bounds it on the refined mesh is that workgroup's memory-level
// parallelism (22 GB/s per CU, HISTORY 8.5), not the chip: at the BASELINE batch of 128 environments half of the CUs idle,
// and 256 environments take 14.2 ms where 128 take 11.3.  Here the two workgroups of a team (the protocol of mode 4: team
// barriers, reductions through the team's slots) share an environment's tile operators: the chunks of a tile application
// are dealt out alternately, every workgroup accumulates its chunks' row sums into ITS OWN vector of the slab (in fixed
// order, as mode 5 does), and behind one team barrier the rows - dealt out over the 1 024 threads of the team like every
// vector pass - add the two partial sums in rank order and run the caller's epilogue.  Bitwise reproducible run to run;
// NOT bitwise mode 5: a row's sum is (chunks 0, 2, 4, ..) + (chunks 1, 3, 5, ..) instead of chunk after chunk.  Without
// tile maps (index data built on the device) the element results go to the slab's element scratch, triangles and rows
// dealt out over the team.  Right-hand sides, the pressure solve (rank 0) and the probes: as in mode 4.
// Which rows a workgroup's chunks touch (t0 / t1: one byte per row) and which entries of its chunks' row lists are the first
// touch of their row in the workgroup's walk (first: one byte per entry) - built once per launch from the row lists: a
// tile application then neither zero-fills the two accumulation vectors nor reads a partial sum that does not exist
// (per application of the refined mesh 1.32 -> 0.57 MB of the ~2.9 MB the workgroups move; the zero fill + full reads were
// what the first version of mode 7 moved more than mode 5, 38.4 against 32.1 GB per step of 128 environments).
struct TeamTouch {
  unsigned char *t0, *t1, *first;   // (first == nullptr: no row lists - zero fill and full reads)
  int stride;                       // entries of `first` per chunk
};
__device__ inline void team_touch_setup(const EnvView& v, const Team& T, TeamTouch& tt) {
  if (!v.mf_rlist || !v.tiles) {
    tt.first = nullptr;
    return;
  }
  const int tid = threadIdx.x, n = v.n2, nch = (v.nt + MF_CH - 1) / MF_CH;
  unsigned char* tm = T.rank ? tt.t1 : tt.t0;
  for (int row = tid; row < n; row += WG) tm[row] = 0;
  __syncthreads();
  for (int chunk = T.rank; chunk < nch; chunk += TEAM) {
    const int2* rl = reinterpret_cast<const int2*>(v.mf_rlist) + (size_t)chunk * v.NRL;
    const int nr = v.mf_rcnt[chunk];
    unsigned char* fbc = tt.first + (size_t)chunk * tt.stride;
    for (int k = tid; k < nr; k += WG) {      // (a row appears once per chunk: its byte is read and set by one thread)
      const int row = rl[k].x & 0x3FFFFFFF;
      fbc[k] = tm[row] == 0;
      tm[row] = 1;
    }
    __syncthreads();
  }
}

template <class ElemOp, class Epi>
__device__ __forceinline__ void tile_apply_team(const EnvView& v, Team& T, bool packed, double2* es, double2* y0v, double2* y1v,
                                                const TeamTouch& tt, const double2* gx, ElemOp op, Epi epi) {
  const int tid = threadIdx.x, n = v.n2, gt = T.rank * WG + tid;
  constexpr int GS = TEAM * WG;
  if (!v.tiles) {
    double2* es2 = reinterpret_cast<double2*>(v.work);
    for (int e = gt; e < v.nt; e += GS) {
      double2 xe[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) xe[i] = gx[v.cell_dofs[i * v.NT + e]];
      const Geo g = load_geo(v, e);
      double2 ye[6];
      op(e, g, (int)v.cell_outflow[e], xe, ye);
#pragma unroll
      for (int i = 0; i < 6; ++i) es2[e * 6 + i] = ye[i];
    }
    team_sync(T);
    for (int row = T.rbeg + tid; row < T.rend; row += WG) {
      double2 a = make_double2(0.0, 0.0);
      for (int s = v.g2_ptr[row]; s < v.g2_ptr[row + 1]; ++s) {
        const double2 c = es2[v.g2_src[s]];
        a.x += c.x;
        a.y += c.y;
      }
      epi(row, a.x, a.y);
    }
    return;       // (the caller's team reduction is the barrier in front of the next use of the element scratch)
  }
  tile_chunks_global(v, packed, es, T.rank ? y1v : y0v, gx, op, T.rank, TEAM, false, tt.first, tt.stride);
  team_sync(T);
  constexpr int EB = 4;
  const bool masks = tt.first != nullptr && v.mf_rlist;
  for (int row0 = T.rbeg + tid; row0 < T.rend; row0 += EB * WG) {
    double2 a[EB], c[EB];
    unsigned m0 = 0xFu, m1 = 0xFu;
    if (masks) {
      // a partial sum exists only where the workgroup's chunks touch the row (mostly ONE of the two: a row meets 1.2 chunks
      // on average) - the other vector's entry is stale and not read
      m0 = m1 = 0u;
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const int rc = min(row0 + k * WG, n - 1);
        m0 |= tt.t0[rc] ? 1u << k : 0u;
        m1 |= tt.t1[rc] ? 1u << k : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < EB; ++k) {
      a[k] = c[k] = make_double2(0.0, 0.0);
      if ((m0 >> k) & 1u) a[k] = y0v[min(row0 + k * WG, n - 1)];
      if ((m1 >> k) & 1u) c[k] = y1v[min(row0 + k * WG, n - 1)];
    }
#pragma unroll
    for (int k = 0; k < EB; ++k)
      if (row0 + k * WG < T.rend) epi(row0 + k * WG, a[k].x + c[k].x, a[k].y + c[k].y);
  }
}

template <bool K1_LDS>
__global__ __launch_bounds__(WG) void evolve_team_tiles_kernel(mdq_ipcs_desc d, int nsteps, double* drag, double* lift,
                                                                int32_t* iters, int general) {
  extern __shared__ __align__(16) double smem[];
  const int q = blockIdx.x / (8 * TEAM), r8 = blockIdx.x % (8 * TEAM);
  const int b = q * 8 + (r8 & 7), rank = r8 >> 3;
  if (b >= d.B || team_test_absent(general, b, rank)) return;
  const int tid = threadIdx.x, gt = rank * WG + tid;
  constexpr int GS = TEAM * WG;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);
  double* red = smem;
  double* U = smem + 64;
  // pressure view of the union (rank 0 only) | velocity view: the element tile (+ the staged rows of a chunk)
  double* px = U;
  double* pr = px + P.NVp;
  double* pp = pr + P.NVp;
  double* pq = pp + P.NVp;
  double* lK = pq + P.NVp;
  int32_t* lci = reinterpret_cast<int32_t*>(lK + d.NSE1);
  int32_t* lso = lci + d.NSE1 + (d.NSE1 & 1);
  double2* es = reinterpret_cast<double2*>(U);
  double* w = v.work;
  double2* escr2 = reinterpret_cast<double2*>(w);
  double* escr1 = w;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);
  double2* vr = xs + d.N2;
  double2* vh = vr + d.N2;
  double2* vp = vh + d.N2;
  double2* vv = vp + d.N2;
  double2* vt = vv + d.N2;
  double2* h1 = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));
  double2* h2 = h1 + d.N2;
  double2* h3 = h2 + d.N2;
  double2* h4 = h3 + d.N2;
  double2* h5 = h4 + d.N2;
  double* hcnt = reinterpret_cast<double*>(h5 + d.N2);
  double2* y0v = reinterpret_cast<double2*>(w + work_ytmp_offset(d.NV, d.NT, d.NE));   // the workgroups' accumulation vectors
  double2* y1v = y0v + d.N2 + 1;
  TeamTouch tt;
  tt.t0 = reinterpret_cast<unsigned char*>(w + work_touch_offset(d.NV, d.NT, d.NE));
  tt.t1 = tt.t0 + d.N2;
  tt.first = tt.t1 + d.N2;
  tt.stride = (int)work_touch_row_cap(d.NV, d.NE);
  double2* stage = vh;                                   // the mass solve's S^-1 p (vh is free there)
  double* pnew = reinterpret_cast<double*>(vt + d.N2);
  double* spare = pnew + d.NV;
  Team T;
  T.rank = rank;
  T.ctr = reinterpret_cast<unsigned*>(spare + 8);
  T.slot = spare + 9;
  T.epoch = 0;
  T.par = 0;
  {
    // the rows of the vector passes and the epilogues: CONTIGUOUS halves (8.45 -> 8.2 ms per step of 128 refined meshes
    // against rows dealt out alternately over the 1 024 threads: the two CUs no longer write into the same lines).  With
    // an odd number of chunks workgroup 0 applies one more, but moving rows to the other workgroup to make up for it was
    // measured slower both ways (0.32 / 0.61 of the rows to workgroup 0: 9.1 / 8.8 ms)
    int split = n2 / 2;
    split &= ~63;
    T.rbeg = rank ? split : 0;
    T.rend = rank ? n2 : split;
  }
  const int nsl1 = (nv + 63) >> 6;
  const int32_t* so1 = K1_LDS ? lso : v.sl1_off;
  const int32_t* ci1 = K1_LDS ? lci : v.sl1_col;
  const double* K1 = K1_LDS ? lK : v.K1s;
  const double a = d.rho / d.dt, mu = d.mu;
  const bool packed = d.N2 <= 4096;
  int it_u = 0, it_p = 0, it_m = 0;
  // y = D^-1 A x (0 on constrained rows) / y = S^-1 M x' on the team's rows
  auto apply_vel = [&](const double2* gx, auto epi) {
    tile_apply_team(
        v, T, packed, es, y0v, y1v, tt, gx,
        [&](int e, const Geo& g, int ko, const double2(&xe)[6], double2(&ye)[6]) { velocity_op(v, a, mu, e, ko, g, xe, ye); },
        [&](int row, double y0, double y1) {
          const bool fl = v.bcu_flag[row] != 0;
          const double2 id = v.idiag1[row];
          epi(row, fl ? 0.0 : y0 * id.x, fl ? 0.0 : y1 * id.y);
        });
  };
  auto apply_mass = [&](const double2* gx, auto epi) {
    tile_apply_team(
        v, T, packed, es, y0v, y1v, tt, gx,
        [&](int, const Geo& g, int, const double2(&xe)[6], double2(&ye)[6]) { elem_mass(g, xe, ye); },
        [&](int row, double y0, double y1) {
          const double is = v.bcu_flag[row] ? 0.0 : 1.0 / v.sdiagM[row];
          epi(row, y0 * is, y1 * is);
        });
  };
  team_place(T, reinterpret_cast<unsigned*>(spare + 25), general);
  if (d.NRL > tt.stride) tt.first = nullptr;       // (cannot happen: a list holds at most min(N2, 6 MF_CH) rows)
  else team_touch_setup(v, T, tt);                  // (published by the team barriers of step 1's right-hand side)
  for (int step = 0; step < nsteps; ++step) {
    // ---------------- step 1: tentative velocity
    for (int e = gt; e < v.nt; e += GS) {
      const ElemIdx E = load_dofs(v, e);
      const Geo g = load_geo(v, e);
      double2 ue[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ue[i] = v.u_n[E.dof[i]];
      double pe[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) pe[i] = v.p_n[E.dof[i]];
      double2 rr_[6];
      elem_rhs1_vol(g, a, mu, d.rho, ue, pe, rr_);
      const int ko = v.cell_outflow[e];
      if (ko >= 0) {
        double X[3][2];
        load_cell_coords(v, e, X);
        elem_outflow_add(g, X, ko, 0.5 * mu, ue, rr_);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) escr2[e * 6 + i] = rr_[i];
    }
    team_sync(T);
    double acc[2] = {0.0, 0.0};
    const int nhist = (int)hcnt[0];
    for (int i = T.rbeg + tid; i < T.rend; i += WG) {
      double2 f = make_double2(0.0, 0.0);
      for (int s_ = v.g2_ptr[i]; s_ < v.g2_ptr[i + 1]; ++s_) {
        const double2 c = escr2[v.g2_src[s_]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift1[i], id = v.idiag1[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) * id.x, (f.y - l.y) * id.y);
      double2 x0 = v.u_n[i];
      if (nhist >= 2) {
        const double2 us1 = h1[i], us2 = h2[i];
        x0 = make_double2(2.0 * us1.x - us2.x, 2.0 * us1.y - us2.y);
        if (nhist >= 3) {
          const double2 us3 = h3[i];
          x0 = make_double2(3.0 * (us1.x - us2.x) + us3.x, 3.0 * (us1.y - us2.y) + us3.y);
          if (nhist >= 4) {
            const double2 us4 = h4[i];
            x0 = make_double2(4.0 * (us1.x + us3.x) - 6.0 * us2.x - us4.x, 4.0 * (us1.y + us3.y) - 6.0 * us2.y - us4.y);
            if (nhist >= 5) {
              const double2 us5 = h5[i];
              x0 = make_double2(5.0 * (us1.x - us4.x) - 10.0 * (us2.x - us3.x) + us5.x,
                                5.0 * (us1.y - us4.y) - 10.0 * (us2.y - us3.y) + us5.y);
            }
          }
        }
      }
      if (fl) x0 = g;
      xs[i] = x0;
      vr[i] = fl ? make_double2(0.0, 0.0) : make_double2(f.x * id.x, f.y * id.y);   // (full operator on x0: no lifting vector)
      acc[0] += bi.x * bi.x + bi.y * bi.y;
    }
    team_sync(T);
    apply_vel(xs, [&](int row, double y0, double y1) {
      const double2 bi = vr[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      vr[row] = r0;
      vh[row] = r0;
      vp[row] = make_double2(0.0, 0.0);
      vv[row] = make_double2(0.0, 0.0);
      acc[1] += r0.x * r0.x + r0.y * r0.y;
    });
    team_sum<2>(acc, red, T);
    {  // BiCGStab (bicgstab_velocity<5>, rows and chunks over the team)
      const double bb = acc[0], tol2 = d.rtol * d.rtol * bb;
      double rr = acc[1];
      if (rr > tol2 && bb != 0.0) {
        double rho = rr, rho_old = 1.0, alpha = 1.0, omega = 1.0;
        int it = 0;
        while (it < d.maxit_u) {
          ++it;
          const double beta = (rho / rho_old) * (alpha / omega);
          for (int i = T.rbeg + tid; i < T.rend; i += WG) {
            const double2 ri = vr[i], pi = vp[i], vi = vv[i];
            vp[i] = make_double2(ri.x + beta * (pi.x - omega * vi.x), ri.y + beta * (pi.y - omega * vi.y));
          }
          team_sync(T);
          double a1[1] = {0.0};
          apply_vel(vp, [&](int row, double y0, double y1) {
            vv[row] = make_double2(y0, y1);
            const double2 h = vh[row];
            a1[0] += h.x * y0 + h.y * y1;
          });
          team_sum<1>(a1, red, T);
          if (a1[0] == 0.0) break;
          alpha = rho / a1[0];
          double a2[1] = {0.0};
          for (int i = T.rbeg + tid; i < T.rend; i += WG) {
            const double2 ri = vr[i], vi = vv[i];
            const double2 sv = make_double2(ri.x - alpha * vi.x, ri.y - alpha * vi.y);
            vr[i] = sv;
            a2[0] += sv.x * sv.x + sv.y * sv.y;
          }
          team_sum<1>(a2, red, T);       // (its team barrier also publishes s)
          if (!(a2[0] > tol2)) {
            for (int i = T.rbeg + tid; i < T.rend; i += WG) {
              const double2 xi = xs[i], pi = vp[i];
              xs[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
            }
            break;
          }
          double a3[2] = {0.0, 0.0};
          apply_vel(vr, [&](int row, double y0, double y1) {
            vt[row] = make_double2(y0, y1);
            const double2 sv = vr[row];
            a3[0] += y0 * sv.x + y1 * sv.y;
            a3[1] += y0 * y0 + y1 * y1;
          });
          team_sum<2>(a3, red, T);
          if (a3[1] == 0.0) break;
          omega = a3[0] / a3[1];
          double a4[2] = {0.0, 0.0};
          for (int i = T.rbeg + tid; i < T.rend; i += WG) {
            const double2 ti = vt[i], xi = xs[i], pi = vp[i], si = vr[i], hi = vh[i];
            xs[i] = make_double2(xi.x + alpha * pi.x + omega * si.x, xi.y + alpha * pi.y + omega * si.y);
            const double2 rn = make_double2(si.x - omega * ti.x, si.y - omega * ti.y);
            vr[i] = rn;
            a4[0] += rn.x * rn.x + rn.y * rn.y;
            a4[1] += hi.x * rn.x + hi.y * rn.y;
          }
          team_sum<2>(a4, red, T);
          rr = a4[0];
          if (!(rr > tol2)) break;
          rho_old = rho;
          rho = a4[1];
          if (rho == 0.0 || omega == 0.0) break;
        }
        it_u += it;
      }
    }
    team_sync(T);
    for (int i = T.rbeg + tid; i < T.rend; i += WG) {  // history, newest first
      if (nhist >= 4) h5[i] = h4[i];
      if (nhist >= 3) h4[i] = h3[i];
      if (nhist >= 2) h3[i] = h2[i];
      if (nhist >= 1) h2[i] = h1[i];
      h1[i] = xs[i];
    }
    // ---------------- step 2: pressure (element loop by the team, the solve by rank 0)
    {
      const double idt = 1.0 / d.dt;
      for (int e = gt; e < v.nt; e += GS) {
        const ElemIdx E = load_dofs(v, e);
        const Geo g = load_geo(v, e);
        double2 ue[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ue[i] = xs[E.dof[i]];
        double pe[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pe[i] = v.p_n[E.dof[i]];
        double r3[3];
        elem_rhs2(g, idt, ue, pe, r3);
#pragma unroll
        for (int j = 0; j < 3; ++j) escr1[e * 3 + j] = r3[j];
      }
    }
    team_sync(T);
    if (rank == 0) {
      if (tid == 0) hcnt[0] = (double)(nhist < 5 ? nhist + 1 : 5);
      if (K1_LDS && !d.pd_enabled) {
        const int ne1 = v.sl1_off[nsl1];
        for (int k = tid; k < ne1; k += WG) {
          lK[k] = v.K1s[k];
          lci[k] = v.sl1_col[k];
        }
        for (int k = tid; k <= nsl1; k += WG) lso[k] = v.sl1_off[k];
      }
      for (int i = tid; i < nv; i += WG) {
        double bsum = 0.0;
        for (int s_ = v.g1_ptr[i]; s_ < v.g1_ptr[i + 1]; ++s_) bsum += escr1[v.g1_src[s_]];
        const double sd = v.sdiagK[i];
        pr[i] = v.bcp_flag[i] ? 0.0 : bsum / sd;
        px[i] = v.p_n[i] * sd;
      }
      if (d.pd_enabled && d.pd_hdr[4 * (int64_t)b + 2] > 0) {
        const PdView pd = pd_view(d, b);
        pressure_direct(pd, nv, pr, px, pp, pq, lK);
      } else {
        it_p += pressure_krylov_lds(d, K1_LDS, nv, so1, ci1, K1, v.coords, px, pr, pp, pq, 2 * sizeof(double) * (size_t)P.NVp, lK,
                                    sizeof(double) * (size_t)P.NVp, red, U,
                                    max(P.prs_vec_bytes + (K1_LDS ? P.prs_mat_bytes : 0), tile_lds_bytes(d)), P.NVp);
      }
      for (int i = tid; i < nv; i += WG) pnew[i] = px[i] / v.sdiagK[i];
    }
    team_sync(T);
    // ---------------- step 3: velocity correction
    for (int e = gt; e < v.nt; e += GS) {
      const ElemIdx E = load_dofs(v, e);
      const Geo g = load_geo(v, e);
      double2 ue[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ue[i] = xs[E.dof[i]];
      double dp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) dp[i] = pnew[E.dof[i]] - v.p_n[E.dof[i]];
      double2 rr_[6];
      elem_rhs3(g, d.dt, ue, dp, rr_);
#pragma unroll
      for (int i = 0; i < 6; ++i) escr2[e * 6 + i] = rr_[i];
    }
    team_sync(T);
    double am[2] = {0.0, 0.0};
    for (int i = T.rbeg + tid; i < T.rend; i += WG) {
      double2 f = make_double2(0.0, 0.0);
      for (int s_ = v.g2_ptr[i]; s_ < v.g2_ptr[i + 1]; ++s_) {
        const double2 c = escr2[v.g2_src[s_]];
        f.x += c.x;
        f.y += c.y;
      }
      const double2 l = v.lift3[i];
      const double sd = v.sdiagM[i];
      const bool fl = v.bcu_flag[i] != 0;
      const double2 g = make_double2(v.bcu_gx[i], 0.0);
      const double2 bi = fl ? g : make_double2((f.x - l.x) / sd, (f.y - l.y) / sd);
      const double2 x0 = fl ? g : xs[i];
      xs[i] = make_double2(x0.x * sd, x0.y * sd);   // scaled unknown S x
      stage[i] = x0;                                 // S^-1 (S x0)
      vr[i] = fl ? make_double2(0.0, 0.0) : make_double2(f.x / sd, f.y / sd);
      am[0] += bi.x * bi.x + bi.y * bi.y;
    }
    team_sync(T);
    apply_mass(stage, [&](int row, double y0, double y1) {
      const double2 bi = vr[row];
      const double2 r0 = make_double2(bi.x - y0, bi.y - y1);
      vr[row] = r0;
      vp[row] = r0;
      am[1] += r0.x * r0.x + r0.y * r0.y;
    });
    team_sum<2>(am, red, T);          // (every gather of x0 from `stage` is behind its team barrier)
    for (int i = T.rbeg + tid; i < T.rend; i += WG) {
      const double is = 1.0 / v.sdiagM[i];
      const double2 p0 = vp[i];
      stage[i] = make_double2(p0.x * is, p0.y * is);
    }
    team_sync(T);
    {  // CG on the mass system (cg_mass<5>)
      const double bb = am[0], tol2 = d.rtol * d.rtol * bb;
      double rr = am[1];
      if (rr > tol2 && bb != 0.0) {
        int it = 0;
        while (it < d.maxit_m) {
          ++it;
          double a1[1] = {0.0};
          apply_mass(stage, [&](int row, double y0, double y1) {
            vv[row] = make_double2(y0, y1);
            const double2 pi = vp[row];
            a1[0] += pi.x * y0 + pi.y * y1;
          });
          team_sum<1>(a1, red, T);
          if (!(a1[0] > 0.0)) break;
          const double alpha = rr / a1[0];
          double a2[1] = {0.0};
          for (int i = T.rbeg + tid; i < T.rend; i += WG) {
            const double2 xi = xs[i], pi = vp[i], ri = vr[i], qi = vv[i];
            xs[i] = make_double2(xi.x + alpha * pi.x, xi.y + alpha * pi.y);
            const double2 rn = make_double2(ri.x - alpha * qi.x, ri.y - alpha * qi.y);
            vr[i] = rn;
            a2[0] += rn.x * rn.x + rn.y * rn.y;
          }
          team_sum<1>(a2, red, T);
          const double rr_new = a2[0];
          if (!(rr_new > tol2)) break;
          const double beta = rr_new / rr;
          rr = rr_new;
          for (int i = T.rbeg + tid; i < T.rend; i += WG) {
            const double2 ri = vr[i], pi = vp[i];
            const double2 pn = make_double2(ri.x + beta * pi.x, ri.y + beta * pi.y);
            vp[i] = pn;
            const double is = 1.0 / v.sdiagM[i];
            stage[i] = make_double2(pn.x * is, pn.y * is);
          }
          team_sync(T);
        }
        it_m += it;
      }
    }
    team_sync(T);
    // ---------------- update state + probes
    if (team_failed(T)) {       // (see team_failed: the state stays the last good one, NaN forces from here on)
      if (rank == 0 && tid == 0) {
        hcnt[0] = 0.0;
        if (d.status) atomicOr(d.status + b, MDQ_IPCS_TEAM_TIMEOUT);
        for (int s_ = step; s_ < nsteps; ++s_) drag[(int64_t)b * nsteps + s_] = lift[(int64_t)b * nsteps + s_] = __builtin_nan("");
      }
      break;
    }
    for (int i = T.rbeg + tid; i < T.rend; i += WG) {
      const double sd = v.sdiagM[i];
      const double2 x = xs[i];
      v.u_n[i] = make_double2(x.x / sd, x.y / sd);
    }
    for (int i = gt; i < nv; i += GS) v.p_n[i] = pnew[i];
    team_sync(T);
    if (rank == 0) {
      double dr, li;
      forces(v, d.mu, v.u_n, v.p_n, red, dr, li);
      if (tid == 0) {
        const bool failed = team_failed(T);
        drag[(int64_t)b * nsteps + step] = failed ? __builtin_nan("") : dr;
        lift[(int64_t)b * nsteps + step] = failed ? __builtin_nan("") : li;
      }
    }
  }
  if (rank == 0 && tid == 0 && iters) {
    iters[3 * b + 0] += it_u;
    iters[3 * b + 1] += it_p;
    iters[3 * b + 2] += it_m;
  }
}

template <bool K1_LDS>
static hipError_t launch_evolve_team_tiles(const mdq_ipcs_desc* d, size_t lds, int nsteps, double* drag, double* lift,
                                           int32_t* iters, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_team_tiles_kernel<K1_LDS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(team_reset_kernel, dim3((d->B + 63) / 64), dim3(64), 0, stream, *d);
  const int teams8 = (d->B + 7) / 8;                       // blocks: groups of 8 environments x TEAM ranks
  hipLaunchKernelGGL((evolve_team_tiles_kernel<K1_LDS>), dim3(teams8 * 8 * TEAM), dim3(WG), lds, stream, *d, nsteps, drag,
                     lift, iters, team_general_barrier());
  return hipGetLastError();
}

// ================================================================== matrix-free kernel, Krylov vectors in registers
//
// MODE 2 proper: every thread keeps its own rows (tid + k*WG, k < MF_ROWS) of ALL Krylov vectors
// in registers for the whole solve; per operator application only the gather copy of the input
// vector goes through LDS (stage) and the element results through the LDS tile.  Global memory is
// touched only for the element metadata (dofs, tile maps, geometry) and once per time step for
// the state vectors.

// float kept in a register -> double at the point of use: the asm stops the compiler from hoisting the conversion
// out of the Krylov loop, which turns 4-byte loop invariants into 8-byte ones (measured: they were then spilled to
// scratch and re-read one by one in every vector phase of every iteration)
__device__ __forceinline__ double f2d(float f) {
  asm volatile("" : "+v"(f));
  return (double)f;
}

// a lane index the compiler cannot see through: the per-row addresses (7 rows x 6 slab vectors x 64 bits) are then
// computed where they are used - hoisted out of the Krylov loop they were what the register allocator spilled
__device__ __forceinline__ int opaque_lane(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

#define MDQ_FOR_ROWS(k, row)                       \
  _Pragma("unroll") for (int k = 0; k < MF_ROWS; ++k) \
    if (const int row = threadIdx.x + k * WG; row < n2)

__device__ __forceinline__ void velocity_op(const EnvView& v, double a, double mu, int e, int ko, const Geo& g,
                                            const double2 (&xe)[6], double2 (&ye)[6]) {
  elem_velocity(g, a, mu, xe, ye);
  if (ko >= 0) {
    double X[3][2];
    load_cell_coords(v, e, X);
    elem_outflow_add(g, X, ko, -0.5 * mu, xe, ye);
  }
}

// Round 5: ONE STEP = THREE LAUNCHES (PHASE 1 tentative velocity, 2 pressure, 3 correction + probes), like mode 3.  The
// persistent kernel of rounds 1-4 (all steps in one launch, r / p / v / t of the BiCGStab solve and x / r / p / q of the mass CG
// in registers across the element phases) sat at 256 VGPRs + 836-932 B per lane of scratch: every operator application
// spilled and re-read ~200 dwords per lane, 428 KB per workgroup, around a phase that needs ~100 registers of its own.  Now
// every phase has its own register allocation, and across an operator application only what the application cannot get
// elsewhere stays in registers: the residual r (+ the fp32 row scaling).  The search direction p is in LDS anyway while it is
// applied (the gather copy `stage`: the owner reads its rows back from there) and, like v = A p and the iterate x, has a
// row-owner-only copy in the environment's workspace slab (coalesced 16-byte accesses that stay in L2): 4 vector passes of
// 53 KB per iteration instead of the scratch traffic.  Same operations on the same values in the same order as before
// (x += alpha p and x += omega s are the two fma's the old single expression contracted to): bitwise reproducible, and the
// element results still meet in the LDS tile in ascending triangle order.
template <bool K1_LDS, int PHASE>
__global__ __launch_bounds__(WG) void evolve_mf_kernel(mdq_ipcs_desc d, int nsteps, int step, double* drag, double* lift,
                                                        int32_t* iters) {
  extern __shared__ __align__(16) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);

  double* red = smem;  // 64 doubles
  double* U = smem + 64;

  double* w = v.work;
  double* escr1 = w;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);  // u* (the iterate of the velocity solve)
  double2* rhg = xs + d.N2;                                            // shadow residual; phase 3: predicted correction, then x
  double* pnew = reinterpret_cast<double*>(xs + 6 * (int64_t)d.N2);
  // initial guesses extrapolated in time: the last five tentative velocities (same slots and counter as evolve_kernel:
  // h1 = newest) and a ring of the last three velocity corrections u_{n+1} - u*.  Every row is read and written by its
  // owner only.
  double2* h1 = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));
  double2* h2 = h1 + d.N2;
  double2* h3 = h2 + d.N2;
  double2* h4 = h3 + d.N2;
  double2* h5 = h4 + d.N2;
  double* hcnt = reinterpret_cast<double*>(h5 + d.N2);   // [0]: tentative velocities stored, [1]: corrections stored
  double2* c1 = xs + 2 * (int64_t)d.N2;                  // newest correction
  double2* c2 = c1 + d.N2;
  double2* c3 = c2 + d.N2;
  double2* pg = c3 + d.N2;                                                              // search direction (own rows)
  double2* vg = reinterpret_cast<double2*>(w + work_ytmp_offset(d.NV, d.NT, d.NE));     // v = D^-1 A p (own rows)

  if constexpr (PHASE == 1) {
    // ================= step 1: tentative velocity
    const double a = d.rho / d.dt, mu = d.mu;
    double2* stage = reinterpret_cast<double2*>(U);  // gather copy of the operator input
    double2* tile = stage + P.N2p;                    // element results of one chunk
    double2* rg = reinterpret_cast<double2*>(w);     // (the element scratch of phase 2, idle here: room for the row scaling)
    TileMeta tm;
    tile_prefetch(v, tm, 0);
    int it_u = 0;
    double2 r[MF_ROWS];
    const int nhist = (int)hcnt[0];
    // Jacobi row scaling of the velocity system with the Dirichlet flag folded in (0 = constrained row), rounded to fp32
    // as in rounds 1-4 (any positive row scaling is a valid left preconditioner and the solution of D^-1 A x = D^-1 b does
    // not depend on it; the same values keep the trajectories of the committed fixtures) - now an own-row vector in the
    // slab like p and v: as a register array it was what the compiler spilled (36 dwords, re-read in every row loop)
    double2* idgg = rg;
#pragma unroll 1
    for (int row = tid; row < n2; row += WG) {
      double2 s_ = make_double2(0.0, 0.0);
      if (!v.bcu_flag[row]) {
        const double2 t_ = v.idiag1[row];
        s_ = make_double2((double)(float)t_.x, (double)(float)t_.y);
      }
      idgg[row] = s_;
    }
    // initial guess: u_n, or the polynomial extrapolation in time of the stored tentative velocities (evolve_kernel's
    // formulas); it satisfies the Dirichlet values
#pragma unroll 1
    for (int row = tid; row < n2; row += WG) {
      double2 x0 = v.u_n[row];
      if (nhist >= 2) {
        const double2 us1 = h1[row], us2 = h2[row];
        x0 = make_double2(2.0 * us1.x - us2.x, 2.0 * us1.y - us2.y);
        if (nhist >= 3) {
          const double2 us3 = h3[row];
          x0 = make_double2(3.0 * (us1.x - us2.x) + us3.x, 3.0 * (us1.y - us2.y) + us3.y);
          if (nhist >= 4) {
            const double2 us4 = h4[row];
            x0 = make_double2(4.0 * (us1.x + us3.x) - 6.0 * us2.x - us4.x, 4.0 * (us1.y + us3.y) - 6.0 * us2.y - us4.y);
            if (nhist >= 5) {
              const double2 us5 = h5[row];
              x0 = make_double2(5.0 * (us1.x - us4.x) - 10.0 * (us2.x - us3.x) + us5.x,
                                5.0 * (us1.y - us4.y) - 10.0 * (us2.y - us3.y) + us5.y);
            }
          }
        }
      }
      if (v.bcu_flag[row] != 0) x0 = make_double2(v.bcu_gx[row], 0.0);
      xs[row] = x0;
      stage[row] = x0;
    }
    double acc[2] = {0.0, 0.0};
    {
      double2 y[MF_ROWS];
      {
        // f = rhs(F1) (volume + outflow facet term), element vectors through the tile
        const double2* un = v.u_n;
        const double* pn = v.p_n;
        tile_accumulate(
            v, tile, tm,
            [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
              double2 ue[6];
#pragma unroll
              for (int i = 0; i < 6; ++i) ue[i] = un[E.dof[i]];
              double pe[3];
#pragma unroll
              for (int i = 0; i < 3; ++i) pe[i] = pn[E.dof[i]];
              elem_rhs1_vol(g, a, mu, d.rho, ue, pe, ye);
              if (ko >= 0) {
                double X[3][2];
                load_cell_coords(v, e, X);
                elem_outflow_add(g, X, ko, 0.5 * mu, ue, ye);
              }
            },
            y);
      }
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        if (row < n2) {
          const bool fl = v.bcu_flag[row] != 0;
          const double2 g = make_double2(v.bcu_gx[row], 0.0);
          // |D^-1 b|^2 with b = f - lift (free) / g (constrained): same norm as the assembled path
          const double2 l = v.lift1[row];
          const double2 sc = idgg[row];
          const double2 bi = fl ? g : make_double2((y[k].x - l.x) * sc.x, (y[k].y - l.y) * sc.y);
          acc[0] += bi.x * bi.x + bi.y * bi.y;
        }
      }
      // (the tile_accumulate above ended with a barrier: the staged x0 of every row is visible)
      double2 ax[MF_ROWS];
      tile_accumulate(
          v, tile, tm,
          [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
            double2 xe[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xe[i] = stage[E.dof[i]];
            velocity_op(v, a, mu, e, ko, g, xe, ye);
          },
          ax);
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        // r0 = D^-1 (f - A_full x0) on free rows, 0 on constrained rows (the scaling is 0 there)
        r[k] = make_double2(0.0, 0.0);
        if (row < n2) {
          const double2 sc = idgg[row];
          r[k] = make_double2((y[k].x - ax[k].x) * sc.x, (y[k].y - ax[k].y) * sc.y);
          rhg[row] = r[k];
          pg[row] = make_double2(0.0, 0.0);
          vg[row] = make_double2(0.0, 0.0);
        }
        acc[1] += r[k].x * r[k].x + r[k].y * r[k].y;
      }
    }
    block_sum<2>(acc, red);
    {
      const double bb = acc[0], tol2 = d.rtol * d.rtol * bb;
      double rr = acc[1];
      if (rr > tol2 && bb != 0.0) {
        double rho = rr, rho_old = 1.0, alpha = 1.0, omega = 1.0;
        int it = 0;
        while (it < d.maxit_u) {
          ++it;
          const double beta = (rho / rho_old) * (alpha / omega);
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = opaque_lane(tid) + k * WG;
            if (row < n2) {
              const double2 po = pg[row], vo = vg[row];
              const double2 pk = make_double2(r[k].x + beta * (po.x - omega * vo.x), r[k].y + beta * (po.y - omega * vo.y));
              pg[row] = pk;
              stage[row] = pk;
            }
          }
          __syncthreads();
          double a1[1] = {0.0};
          double2 vv[MF_ROWS];                   // (live until s = r - alpha v below; its slab copy is for the next p update)
          {
            tile_accumulate(
                v, tile, tm,
                [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
                  double2 xe[6];
#pragma unroll
                  for (int i = 0; i < 6; ++i) xe[i] = stage[E.dof[i]];
                  velocity_op(v, a, mu, e, ko, g, xe, ye);
                },
                vv);
#pragma unroll
            for (int k = 0; k < MF_ROWS; ++k) {
              const int row = opaque_lane(tid) + k * WG;
              if (row < n2) {
                const double2 h = rhg[row], sc = idgg[row];
                vv[k] = make_double2(vv[k].x * sc.x, vv[k].y * sc.y);
                a1[0] += h.x * vv[k].x + h.y * vv[k].y;
                vg[row] = vv[k];
              }
            }
          }
          block_sum<1>(a1, red);
          if (a1[0] == 0.0) break;
          alpha = rho / a1[0];
          double a2[1] = {0.0};
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = opaque_lane(tid) + k * WG;
            if (row < n2) {
              const double2 vo = vv[k], pk = stage[row], xo = xs[row];    // (own rows)
              r[k] = make_double2(r[k].x - alpha * vo.x, r[k].y - alpha * vo.y);  // s
              xs[row] = make_double2(xo.x + alpha * pk.x, xo.y + alpha * pk.y);
              stage[row] = r[k];
            }
            else
              r[k] = make_double2(0.0, 0.0);
            a2[0] += r[k].x * r[k].x + r[k].y * r[k].y;
          }
          block_sum<1>(a2, red);  // its barriers publish the staged s
          if (!(a2[0] > tol2)) break;            // (x = x + alpha p is already in place)
          double a3[2] = {0.0, 0.0};
          double2 t[MF_ROWS];
          tile_accumulate(
              v, tile, tm,
              [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
                double2 xe[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) xe[i] = stage[E.dof[i]];
                velocity_op(v, a, mu, e, ko, g, xe, ye);
              },
              t);
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = opaque_lane(tid) + k * WG;
            const double2 sc = row < n2 ? idgg[row] : make_double2(0.0, 0.0);
            t[k] = make_double2(t[k].x * sc.x, t[k].y * sc.y);
            a3[0] += t[k].x * r[k].x + t[k].y * r[k].y;
            a3[1] += t[k].x * t[k].x + t[k].y * t[k].y;
          }
          block_sum<2>(a3, red);
          if (a3[1] == 0.0) break;
          omega = a3[0] / a3[1];
          double a4[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = opaque_lane(tid) + k * WG;
            if (row < n2) {
              const double2 xo = xs[row], h = rhg[row];
              xs[row] = make_double2(xo.x + omega * r[k].x, xo.y + omega * r[k].y);
              r[k] = make_double2(r[k].x - omega * t[k].x, r[k].y - omega * t[k].y);
              a4[0] += r[k].x * r[k].x + r[k].y * r[k].y;
              a4[1] += h.x * r[k].x + h.y * r[k].y;
            }
          }
          block_sum<2>(a4, red);
          rr = a4[0];
          if (!(rr > tol2)) break;
          rho_old = rho;
          rho = a4[1];
          if (rho == 0.0 || omega == 0.0) break;
        }
        it_u += it;
      }
    }
#pragma unroll 1
    for (int row = tid; row < n2; row += WG) {   // shift the history, newest first: h1 = u* of this step (own rows)
      if (nhist >= 4) h5[row] = h4[row];
      if (nhist >= 3) h4[row] = h3[row];
      if (nhist >= 2) h3[row] = h2[row];
      if (nhist >= 1) h2[row] = h1[row];
      h1[row] = xs[row];
    }
    if (tid == 0) hcnt[0] = (double)(nhist < 5 ? nhist + 1 : 5);
    if (tid == 0 && iters) iters[3 * b + 0] += it_u;
  } else if constexpr (PHASE == 2) {
    // ================= step 2: pressure (assembled K1 in SELL form, LDS resident)
    double* px = U;
    double* pr = px + P.NVp;
    double* pp = pr + P.NVp;
    double* pq = pp + P.NVp;
    double* lK = pq + P.NVp;
    int32_t* lci = reinterpret_cast<int32_t*>(lK + d.NSE1);
    int32_t* lso = lci + d.NSE1 + (d.NSE1 & 1);
    const int nsl1 = (nv + 63) >> 6;
    const int32_t* so1 = K1_LDS ? lso : v.sl1_off;
    const int32_t* ci1 = K1_LDS ? lci : v.sl1_col;
    const double* K1 = K1_LDS ? lK : v.K1s;
    int it_p = 0;
    if (K1_LDS && !d.pd_enabled) {
      const int ne1 = v.sl1_off[nsl1];
      for (int kk = tid; kk < ne1; kk += WG) {
        lK[kk] = v.K1s[kk];
        lci[kk] = v.sl1_col[kk];
      }
      for (int kk = tid; kk <= nsl1; kk += WG) lso[kk] = v.sl1_off[kk];
    }
    rhs2_elements(v, d, xs, v.p_n, escr1);
    __syncthreads();
    for (int i = tid; i < nv; i += WG) {
      double bsum = 0.0;
      for (int s = v.g1_ptr[i]; s < v.g1_ptr[i + 1]; ++s) bsum += escr1[v.g1_src[s]];
      const double sd = v.sdiagK[i];
      pr[i] = v.bcp_flag[i] ? 0.0 : bsum / sd;
      px[i] = v.p_n[i] * sd;
    }
    if (d.pd_enabled && d.pd_hdr[4 * (int64_t)b + 2] > 0) {   // (nparts = 0: no factors for this environment -> Krylov)
      const PdView pd = pd_view(d, b);
      pressure_direct(pd, nv, pr, px, pp, pq, lK);
    } else {
      it_p += cg_pressure(nv, so1, ci1, K1, d.rtol, d.maxit_p, px, pr, pp, pq, red);
    }
    for (int i = tid; i < nv; i += WG) pnew[i] = px[i] / v.sdiagK[i];
    if (tid == 0 && iters) iters[3 * b + 1] += it_p;
  } else {
    // ================= step 3: velocity correction (mass solve, both components)
    double2* stage = reinterpret_cast<double2*>(U);
    double2* tile = stage + P.N2p;
    double2* xg = rhg;              // the scaled unknown S x of the mass CG (own rows), once the predicted correction is consumed
    TileMeta tm;
    tile_prefetch(v, tm, 0);
    int it_m = 0;
    const int ncorr = (int)hcnt[1];
    double am[2] = {0.0, 0.0};
    double2 r[MF_ROWS], p[MF_ROWS];
    // symmetric Jacobi scaling S^-1 of the mass system, 0 on constrained rows
    double ism[MF_ROWS];
    {
      double2 y[MF_ROWS];
      {
        const double* pold = v.p_n;
        tile_accumulate(
            v, tile, tm,
            [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
              double2 ue[6];
#pragma unroll
              for (int i = 0; i < 6; ++i) ue[i] = xs[E.dof[i]];
              double dp[3];
#pragma unroll
              for (int i = 0; i < 3; ++i) dp[i] = pnew[E.dof[i]] - pold[E.dof[i]];
              elem_rhs3(g, d.dt, ue, dp, ye);
            },
            y);
      }
      double2 x[MF_ROWS];
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        // u* (own rows) + the correction extrapolated from the ring (0 on constrained rows: the corrections vanish there)
        x[k] = make_double2(0.0, 0.0);
        ism[k] = 0.0;
        if (row < n2) {
          double2 dp_ = make_double2(0.0, 0.0);
          if (ncorr >= 1) {
            const double2 d1 = c1[row];
            dp_ = d1;
            if (ncorr >= 2) {
              const double2 d2 = c2[row];
              dp_ = make_double2(2.0 * d1.x - d2.x, 2.0 * d1.y - d2.y);
              if (ncorr >= 3) {
                const double2 d3 = c3[row];
                dp_ = make_double2(3.0 * (d1.x - d2.x) + d3.x, 3.0 * (d1.y - d2.y) + d3.y);
              }
            }
          }
          const double2 us = xs[row];
          x[k] = make_double2(us.x + dp_.x, us.y + dp_.y);
          // x satisfies the Dirichlet values: scaled unknown S x, stage S^-1 (S x0) = x0
          const bool fl = v.bcu_flag[row] != 0;
          if (!fl) ism[k] = 1.0 / v.sdiagM[row];
          stage[row] = x[k];
          const double2 l = v.lift3[row];
          const double2 bi = fl ? x[k] : make_double2((y[k].x - l.x) * ism[k], (y[k].y - l.y) * ism[k]);
          am[0] += bi.x * bi.x + bi.y * bi.y;
        }
      }
      __syncthreads();
      double2 ax[MF_ROWS];
      tile_accumulate(
          v, tile, tm,
          [&](int, const Geo& g, const ElemIdx& E, int, double2(&ye)[6]) {
            double2 xe[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xe[i] = stage[E.dof[i]];
            elem_mass(g, xe, ye);
          },
          ax);
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        r[k] = make_double2((y[k].x - ax[k].x) * ism[k], (y[k].y - ax[k].y) * ism[k]);  // 0 on constrained rows
        p[k] = r[k];
        am[1] += r[k].x * r[k].x + r[k].y * r[k].y;
        // scaled unknown: S x (S = 1/ism on free rows; constrained rows keep x = g with S = 1)
        if (ism[k] != 0.0) x[k] = make_double2(x[k].x / ism[k], x[k].y / ism[k]);
        if (row < n2) xg[row] = x[k];
      }
    }
    block_sum<2>(am, red);
    {
      const double bb = am[0], tol2 = d.rtol * d.rtol * bb;
      double rr = am[1];
      if (rr > tol2 && bb != 0.0) {
        int it = 0;
        while (it < d.maxit_m) {
          ++it;
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = tid + k * WG;
            if (row < n2) stage[row] = make_double2(p[k].x * ism[k], p[k].y * ism[k]);
          }
          __syncthreads();
          double2 q[MF_ROWS];
          tile_accumulate(
              v, tile, tm,
              [&](int, const Geo& g, const ElemIdx& E, int, double2(&ye)[6]) {
                double2 xe[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) xe[i] = stage[E.dof[i]];
                elem_mass(g, xe, ye);
              },
              q);
          double a1[1] = {0.0};
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            q[k] = make_double2(q[k].x * ism[k], q[k].y * ism[k]);
            a1[0] += p[k].x * q[k].x + p[k].y * q[k].y;
          }
          block_sum<1>(a1, red);
          if (!(a1[0] > 0.0)) break;
          const double alpha = rr / a1[0];
          double a2[1] = {0.0};
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = tid + k * WG;
            if (row < n2) {
              const double2 xo = xg[row];
              xg[row] = make_double2(xo.x + alpha * p[k].x, xo.y + alpha * p[k].y);
            }
            r[k] = make_double2(r[k].x - alpha * q[k].x, r[k].y - alpha * q[k].y);
            a2[0] += r[k].x * r[k].x + r[k].y * r[k].y;
          }
          block_sum<1>(a2, red);
          const double rr_new = a2[0];
          if (!(rr_new > tol2)) break;
          const double beta = rr_new / rr;
          rr = rr_new;
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k)
            p[k] = make_double2(r[k].x + beta * p[k].x, r[k].y + beta * p[k].y);
        }
        it_m += it;
      }
    }

    // ================= update state + probes
#pragma unroll
    for (int k = 0; k < MF_ROWS; ++k) {
      const int row = tid + k * WG;
      if (row < n2) {
        const double2 xk = xg[row], us = xs[row];
        const double2 un = (ism[k] != 0.0) ? make_double2(xk.x * ism[k], xk.y * ism[k]) : xk;
        v.u_n[row] = un;
        // ring of corrections, newest first (own rows)
        if (ncorr >= 2) c3[row] = c2[row];
        if (ncorr >= 1) c2[row] = c1[row];
        c1[row] = make_double2(un.x - us.x, un.y - us.y);
      }
    }
    if (tid == 0) hcnt[1] = (double)(ncorr < 3 ? ncorr + 1 : 3);
    for (int i = tid; i < nv; i += WG) v.p_n[i] = pnew[i];
    __syncthreads();
    double dr, li;
    forces(v, d.mu, v.u_n, v.p_n, red, dr, li);
    if (tid == 0) {
      drag[(int64_t)b * nsteps + step] = dr;
      lift[(int64_t)b * nsteps + step] = li;
    }
    if (tid == 0 && iters) iters[3 * b + 2] += it_m;
  }
}

template <bool K1_LDS>
static hipError_t launch_evolve_mf(const mdq_ipcs_desc* d, size_t lds, int nsteps, double* drag, double* lift,
                                   int32_t* iters, hipStream_t stream) {
  static const hipError_t attr = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_mf_kernel<K1_LDS, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_mf_kernel<K1_LDS, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&evolve_mf_kernel<K1_LDS, 3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
  }();
  if (attr != hipSuccess) return attr;
  const LdsPlan P = lds_plan(d->N2, d->NV, d->NSE1);
  const size_t red_bytes = 64 * sizeof(double);
  const size_t lds_v = red_bytes + P.vel2_bytes;
  const size_t lds_p = red_bytes + P.prs_vec_bytes + (K1_LDS ? P.prs_mat_bytes : 0);
  (void)lds;
  for (int step = 0; step < nsteps; ++step) {
    hipLaunchKernelGGL((evolve_mf_kernel<K1_LDS, 1>), dim3(d->B), dim3(WG), lds_v, stream, *d, nsteps, step, drag, lift, iters);
    hipLaunchKernelGGL((evolve_mf_kernel<K1_LDS, 2>), dim3(d->B), dim3(WG), lds_p, stream, *d, nsteps, step, drag, lift, iters);
    hipLaunchKernelGGL((evolve_mf_kernel<K1_LDS, 3>), dim3(d->B), dim3(WG), lds_v, stream, *d, nsteps, step, drag, lift, iters);
  }
  return hipGetLastError();
}

// ================================================================== matrix-free kernel, LDS-atomic accumulation (mode 3)
//
// Same element operators as mode 2, but the element results are accumulated straight into an LDS
// result vector with hardware LDS fp64 atomics (ds_add_f64) instead of going through a tile and an
// ordered row gather.  That frees the 96 KB tile: the search direction p, the residual r (= s) AND
// the result vector all live in LDS, only v stays in registers, so the element phase can interleave
// several triangles per thread (FP64 ILP) without register spills, with two barriers per operator
// application.  Price: the summation order of the <= 8 contributions per row depends on timing, so
// results are reproducible to round-off only (modes 0-2 are bitwise reproducible).

constexpr int AT_PAIR = 2;  // triangles interleaved per thread per round

template <int TPAIR = AT_PAIR>
struct AtMetaT {
  int w[TPAIR][6];
  Geo g[TPAIR];
};
using AtMeta = AtMetaT<AT_PAIR>;

template <int TW = WG, int TPAIR = AT_PAIR>
__device__ __forceinline__ void at_prefetch(const EnvView& v, AtMetaT<TPAIR>& m, int round) {
#pragma unroll
  for (int j = 0; j < TPAIR; ++j) {
    const int e = threadIdx.x + (round * TPAIR + j) * TW;
    if (e < v.nt) {
#pragma unroll
      for (int i = 0; i < 6; ++i) m.w[j][i] = v.mf_scat[i * v.NT + e];
      m.g[j] = load_geo(v, e);
    }
  }
}

// Y[dof] += element results for every triangle (Y zeroed and published by the caller).
// op(e, geo, dofs, outflow_edge, ye).  On entry m holds round 0; on exit again (rolling prefetch).
// INTERLEAVE = true computes the AT_PAIR triangles of a round side by side (FP64 ILP for the hot
// operator applications); false runs them one after the other (register-hungry right-hand sides).
template <bool INTERLEAVE, int TW = WG, int TPAIR = AT_PAIR, class ElemOp>
__device__ __forceinline__ void atomic_accumulate(const EnvView& v, double* Yd, AtMetaT<TPAIR>& m, ElemOp op) {
  const int nrounds = (v.nt + TPAIR * TW - 1) / (TPAIR * TW);
  // (delaying the odd waves by 256-1536 cycles to de-phase LDS and FP64 phases only added the delay: measured)
  for (int round = 0; round < nrounds; ++round) {
    if (INTERLEAVE) {
      double2 ye[TPAIR][6];
      int dof[TPAIR][6];
#pragma unroll
      for (int j = 0; j < TPAIR; ++j) {
        const int e = threadIdx.x + (round * TPAIR + j) * TW;
        if (e < v.nt) {
          ElemIdx E;
#pragma unroll
          for (int i = 0; i < 6; ++i) E.dof[i] = dof[j][i] = m.w[j][i] & 0xFFF;
          op(e, m.g[j], E, ((m.w[j][0] >> 28) & 3) - 1, ye[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < TPAIR; ++j) {
        const int e = threadIdx.x + (round * TPAIR + j) * TW;
        if (e < v.nt) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
#ifdef MDQ_AT_U64HACK   // timing experiment only (wrong numerics): 64-bit integer atomics instead of fp64 ones
            atomicAdd(reinterpret_cast<unsigned long long*>(Yd) + 2 * dof[j][i], (unsigned long long)__double_as_longlong(ye[j][i].x) >> 40);
            atomicAdd(reinterpret_cast<unsigned long long*>(Yd) + 2 * dof[j][i] + 1, (unsigned long long)__double_as_longlong(ye[j][i].y) >> 40);
#else
            unsafeAtomicAdd(Yd + 2 * dof[j][i], ye[j][i].x);
            unsafeAtomicAdd(Yd + 2 * dof[j][i] + 1, ye[j][i].y);
#endif
          }
        }
      }
    } else {
#pragma unroll 1
      for (int j = 0; j < TPAIR; ++j) {
        const int e = threadIdx.x + (round * TPAIR + j) * TW;
        if (e < v.nt) {
          ElemIdx E;
          double2 ye[6];
          // (explicit selects: a runtime index into the register arrays would demote them to scratch)
          const int* wj = (TPAIR == 1 || j == 0) ? m.w[0] : m.w[TPAIR - 1];
          const Geo gj = (TPAIR == 1 || j == 0) ? m.g[0] : m.g[TPAIR - 1];
#pragma unroll
          for (int i = 0; i < 6; ++i) E.dof[i] = wj[i] & 0xFFF;
          op(e, gj, E, ((wj[0] >> 28) & 3) - 1, ye);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
#ifdef MDQ_AT_U64HACK
            atomicAdd(reinterpret_cast<unsigned long long*>(Yd) + 2 * E.dof[i], (unsigned long long)__double_as_longlong(ye[i].x) >> 40);
            atomicAdd(reinterpret_cast<unsigned long long*>(Yd) + 2 * E.dof[i] + 1, (unsigned long long)__double_as_longlong(ye[i].y) >> 40);
#else
            unsafeAtomicAdd(Yd + 2 * E.dof[i], ye[i].x);
            unsafeAtomicAdd(Yd + 2 * E.dof[i] + 1, ye[i].y);
#endif
          }
        }
      }
    }
    at_prefetch<TW, TPAIR>(v, m, round + 1 < nrounds ? round + 1 : 0);
  }
}

// Outflow rows owned by this thread (row % WG == tid): at most BO_OWN of them, found once per launch.
constexpr int BO_OWN = 2;
struct BoOwn {
  int t[BO_OWN];
};
template <int TW = WG>
__device__ __forceinline__ BoOwn outflow_rows_owned(const EnvView& v) {
  BoOwn o;
#pragma unroll
  for (int q = 0; q < BO_OWN; ++q) o.t[q] = -1;
  for (int t = 0; t < v.nbo; ++t) {
    if ((v.bo_rows[t] % TW) == (int)threadIdx.x) {
      if (o.t[0] < 0) o.t[0] = t;
      else o.t[1] = t;  // (a third owned row cannot occur: outflow rows are < 2*WG apart in practice; checked on the host)
    }
  }
  return o;
}

// Y[row] += coef * sum_entries B x[col] for the outflow rows owned by this thread.
// Called between the barrier that completes the atomic accumulation and the owner's read of Y;
// x must not be modified by other threads before the next barrier.
__device__ __forceinline__ void outflow_rows_add(const EnvView& v, const BoOwn& own, double coef, const double2* x,
                                                 double2* Y) {
#pragma unroll
  for (int q = 0; q < BO_OWN; ++q) {
    const int t = own.t[q];
    if (t >= 0) {
      const int row = v.bo_rows[t];
      double sx = 0.0, sy = 0.0;
      for (int k = v.bo_ptr[t]; k < v.bo_ptr[t + 1]; ++k) {
        const double4 bv = reinterpret_cast<const double4*>(v.bo_val)[k];
        const double2 xc = x[v.bo_col[k]];
        sx += bv.x * xc.x + bv.y * xc.y;
        sy += bv.z * xc.x + bv.w * xc.y;
      }
      const double2 y = Y[row];
      Y[row] = make_double2(y.x + coef * sx, y.y + coef * sy);
    }
  }
}


// The same outflow-facet term, one ENTRY per thread (18 entries per outflow facet, a few hundred per mesh): every
// thread adds coef * B_entry x[col] to Y[row] with LDS atomics, between the barrier that publishes the zeroed Y
// and the element loop, so that the term costs one global-load latency of the waves that own entries instead of a
// serial per-row chain of them between two barriers.  (row, col) of entry `tid` is found once per launch from the
// element slot list (bo_src = cell * 36 + i * 6 + j) and kept packed in one register; entries beyond the
// workgroup size (never on the training meshes) are looked up on the fly.
struct BoEnt {
  int rc;   // row << 12 | col of entry threadIdx.x, -1 if none
  int nbe;  // entries of this environment
};
__device__ __forceinline__ int outflow_entry_rc(const EnvView& v, int t) {
  const int slot = v.bo_src[t];
  const int e = slot / 36, ij = slot - e * 36, i = ij / 6, j = ij - i * 6;
  return ((v.mf_scat[i * v.NT + e] & 0xFFF) << 12) | (v.mf_scat[j * v.NT + e] & 0xFFF);
}
__device__ __forceinline__ BoEnt outflow_entries_owned(const EnvView& v) {
  BoEnt o;
  o.nbe = v.nbo > 0 ? v.bo_ptr[v.nbo] : 0;
  o.rc = (int)threadIdx.x < o.nbe ? outflow_entry_rc(v, threadIdx.x) : -1;
  return o;
}
__device__ __forceinline__ void outflow_entry_apply(const EnvView& v, int t, int rc, double coef, const double2* x,
                                                    double* Yd) {
  const double4 bv = reinterpret_cast<const double4*>(v.bo_val)[t];
  const double2 xc = x[rc & 0xFFF];
  const int row = rc >> 12;
  unsafeAtomicAdd(Yd + 2 * row, coef * (bv.x * xc.x + bv.y * xc.y));
  unsafeAtomicAdd(Yd + 2 * row + 1, coef * (bv.z * xc.x + bv.w * xc.y));
}
template <int TW>
__device__ __forceinline__ void outflow_entries_add(const EnvView& v, const BoEnt& o, double coef, const double2* x,
                                                    double* Yd) {
  if (o.rc >= 0) outflow_entry_apply(v, threadIdx.x, o.rc, coef, x, Yd);
  for (int t = threadIdx.x + TW; t < o.nbe; t += TW) outflow_entry_apply(v, t, outflow_entry_rc(v, t), coef, x, Yd);
}


// a workgroup-uniform double moved to scalar registers (two v_readfirstlane): loop invariants such as rho / dt, mu or the
// squared tolerance otherwise occupy a VGPR pair each for the whole Krylov loop (the 768-thread velocity kernel, capped at
// 168 VGPRs, spilled two of them and re-read them in every iteration)
__device__ __forceinline__ double uniform_double(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- mode 3 as three kernels per time step (separate register allocation per phase: the BiCGStab
// loop then runs without spill reloads; state is handed over through global memory as before) ----
template <int TW, int TROWS, int TPAIR>
__global__ __launch_bounds__(TW) void at_velocity_kernel(mdq_ipcs_desc d, int32_t* iters) {
  extern __shared__ __align__(16) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);
  const double a = uniform_double(d.rho / d.dt), mu = d.mu;
  (void)a; (void)mu; (void)nv; (void)n2;
  double* red = smem;  // 64 doubles
  double* U = smem + 64;
  double* w = v.work;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);  // x of the velocity solve = u*
  double* pnew = reinterpret_cast<double*>(xs + 6 * (int64_t)d.N2);
  int rsel = 0;  // parity of the one-barrier reductions
  (void)rsel; (void)pnew; (void)red;
  double2* Pl = reinterpret_cast<double2*>(U);  // search direction p / staged operator input
  double2* Rl = Pl + P.N2p;                      // residual r (= s)
  double2* Yl = Rl + P.N2p;                      // operator result (atomic accumulation)
  double* Yd = reinterpret_cast<double*>(Yl);
  AtMetaT<TPAIR> tm;
  at_prefetch<TW, TPAIR>(v, tm, 0);
  const BoEnt bo_ent = outflow_entries_owned(v);
  int it_u = 0;
  double2* hist = xs + d.N2;                                          // u* of the step before the last
  double2* hist2 = xs + 2 * (int64_t)d.N2;                            // and of the one before that
  double2* hist3 = xs + 4 * (int64_t)d.N2;                            // and one more (slot 3 holds the counter)
  double2* hist4 = xs + 5 * (int64_t)d.N2;
  double* histc = reinterpret_cast<double*>(xs + 3 * (int64_t)d.N2);  // [0]: tentative velocities stored so far
  const int nhist = (int)histc[0];
#ifdef MDQ_AT_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  AT_STAMP(0)
  {
    // ================= step 1: tentative velocity
    // Row loops: thread `tid` owns rows tid + k * TW.  All reads use the clamped row (rows past n2 re-read row
    // n2 - 1, value unused) and come first, so that the TROWS loads of a phase are in flight together instead of
    // one conditional block (and one LDS / L2 round trip) per row; only the writes are predicated.
    int rc[TROWS];
    float2 idg[TROWS];
    unsigned flm = 0;  // bit k: row k of this thread carries a Dirichlet value
#pragma unroll
    for (int k = 0; k < TROWS; ++k) rc[k] = min(tid + k * TW, n2 - 1);
    {
      // u_n and p_n staged in LDS (p and r are free until the solve starts): the right-hand-side element loop
      // gathers 6 + 3 values per triangle from LDS instead of from L2
      double2 un_[TROWS], dg_[TROWS];
      unsigned char fl_[TROWS];
#pragma unroll
      for (int k = 0; k < TROWS; ++k) {
        un_[k] = v.u_n[rc[k]];
        dg_[k] = v.idiag1[rc[k]];
        fl_[k] = v.bcu_flag[rc[k]];
      }
      double* Pn = reinterpret_cast<double*>(Rl);
      for (int i = tid; i < nv; i += TW) Pn[i] = v.p_n[i];
#pragma unroll
      for (int k = 0; k < TROWS; ++k) {
        const int row = tid + k * TW;
        idg[k] = (row < n2 && !fl_[k]) ? make_float2((float)dg_[k].x, (float)dg_[k].y) : make_float2(0.f, 0.f);
        flm |= fl_[k] ? 1u << k : 0u;
        if (row < n2) {
          Pl[row] = un_[k];
          Yl[row] = make_double2(0.0, 0.0);
        }
      }
    }
    __syncthreads();
    outflow_entries_add<TW>(v, bo_ent, 0.5 * mu, Pl, Yd);  // + mu/2 <nabla_grad(u_n) n, v> on the outflow rows
    {
      const double* Pn = reinterpret_cast<const double*>(Rl);
      atomic_accumulate<false, TW, TPAIR>(v, Yd, tm, [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
        double2 ue[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ue[i] = Pl[E.dof[i]];
        double pe[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pe[i] = Pn[E.dof[i]];
        elem_rhs1_vol(g, a, mu, d.rho, ue, pe, ye);
      });
    }
    // initial guess: polynomial extrapolation in time of the previous tentative velocities (xs still holds
    // u*_n, hist.. hist4 the four before it), up to quartic as the history fills; u_n while there is none.
    // dt is far below the flow's time scales, so every order removes about two digits of initial residual
    // (11 BiCGStab iterations without, 8 linear, 6 quadratic, 3 cubic, 2.6 quartic; beyond that the 1e-10
    // solver noise of the history, amplified by the coefficients, is the floor).  Dirichlet values hold.
    // The history loads are issued here, in front of the barrier that ends the element loop.
    double2 x0[TROWS];
    {
      // binomial coefficients of the order in use: x0 = c1 u*_n + c2 u*_{n-1} + ... (alternating signs)
      const double c1 = nhist >= 5 ? 5.0 : nhist == 4 ? 4.0 : nhist == 3 ? 3.0 : 2.0;
      const double c2 = nhist >= 5 ? -10.0 : nhist == 4 ? -6.0 : nhist == 3 ? -3.0 : -1.0;
      const double c3 = nhist >= 5 ? 10.0 : nhist == 4 ? 4.0 : 1.0;
      const double c4 = nhist >= 5 ? -5.0 : -1.0;
      double2 us1[TROWS];
#pragma unroll
      for (int k = 0; k < TROWS; ++k) us1[k] = xs[rc[k]];
      if (nhist >= 2) {
        double2 h[TROWS];
#pragma unroll
        for (int k = 0; k < TROWS; ++k) h[k] = hist[rc[k]];
#pragma unroll
        for (int k = 0; k < TROWS; ++k) x0[k] = make_double2(c1 * us1[k].x + c2 * h[k].x, c1 * us1[k].y + c2 * h[k].y);
        if (nhist >= 3) {
          double2 h2[TROWS];
#pragma unroll
          for (int k = 0; k < TROWS; ++k) h2[k] = hist2[rc[k]];
#pragma unroll
          for (int k = 0; k < TROWS; ++k) x0[k] = make_double2(x0[k].x + c3 * h2[k].x, x0[k].y + c3 * h2[k].y);
          if (nhist >= 4) {
            double2 h3[TROWS];
#pragma unroll
            for (int k = 0; k < TROWS; ++k) h3[k] = hist3[rc[k]];
#pragma unroll
            for (int k = 0; k < TROWS; ++k) x0[k] = make_double2(x0[k].x + c4 * h3[k].x, x0[k].y + c4 * h3[k].y);
            if (nhist >= 5) {
#pragma unroll
              for (int k = 0; k < TROWS; ++k) {
                const double2 h4 = hist4[rc[k]];
                x0[k] = make_double2(x0[k].x + h4.x, x0[k].y + h4.y);
              }
            }
#pragma unroll
            for (int k = 0; k < TROWS; ++k)
              if (tid + k * TW < n2) hist4[tid + k * TW] = h3[k];
          }
#pragma unroll
          for (int k = 0; k < TROWS; ++k)
            if (tid + k * TW < n2) hist3[tid + k * TW] = h2[k];
        }
#pragma unroll
        for (int k = 0; k < TROWS; ++k)
          if (tid + k * TW < n2) hist2[tid + k * TW] = h[k];
      } else {
#pragma unroll
        for (int k = 0; k < TROWS; ++k) x0[k] = v.u_n[rc[k]];
      }
#pragma unroll
      for (int k = 0; k < TROWS; ++k)
        if (tid + k * TW < n2) hist[tid + k * TW] = us1[k];
    }
    double2 lf[TROWS];
    double gx[TROWS];
#pragma unroll
    for (int k = 0; k < TROWS; ++k) {
      lf[k] = v.lift1[rc[k]];
      gx[k] = v.bcu_gx[rc[k]];
    }
    __syncthreads();
    AT_STAMP(1)
    double acc[2] = {0.0, 0.0};
    double2 f[TROWS];
#pragma unroll
    for (int k = 0; k < TROWS; ++k) f[k] = Yl[rc[k]];
#pragma unroll
    for (int k = 0; k < TROWS; ++k) {
      const int row = tid + k * TW;
      const bool fl = (flm >> k) & 1u;
      const double2 g = make_double2(gx[k], 0.0);
      if (fl) x0[k] = g;
      if (row < n2) {
        Yl[row] = make_double2(0.0, 0.0);
        xs[row] = x0[k];
        Pl[row] = x0[k];
        const double2 bi = fl ? g : make_double2((f[k].x - lf[k].x) * f2d(idg[k].x), (f[k].y - lf[k].y) * f2d(idg[k].y));
        acc[0] += bi.x * bi.x + bi.y * bi.y;
      }
    }
    __syncthreads();
    AT_STAMP(2)
    outflow_entries_add<TW>(v, bo_ent, -0.5 * mu, Pl, Yd);
    atomic_accumulate<(TPAIR > 1), TW, TPAIR>(v, Yd, tm, [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
      double2 xe[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) xe[i] = Pl[E.dof[i]];
      elem_velocity(g, a, mu, xe, ye);
    });
    __syncthreads();
    AT_STAMP(3)
    double2 vv[TROWS];
#pragma unroll
    for (int k = 0; k < TROWS; ++k) {
      const int row = tid + k * TW;
      vv[k] = make_double2(0.0, 0.0);
      if (row < n2) {
        const double2 ax = Yl[row];
        // r0 = D^-1 (f - A_full x0) on free rows, 0 on constrained rows (idg = 0 there)
        const double2 r0 = make_double2((f[k].x - ax.x) * f2d(idg[k].x), (f[k].y - ax.y) * f2d(idg[k].y));
        Rl[row] = r0;
        acc[1] += r0.x * r0.x + r0.y * r0.y;
      }
    }
    block_sum<2, TW / 64>(acc, red);
    // p = 0 (only now: the outflow rows above still gathered x0 from Pl until the barriers of the reduction)
#pragma unroll
    for (int k = 0; k < TROWS; ++k) {
      const int row = tid + k * TW;
      if (row < n2) Pl[row] = make_double2(0.0, 0.0);
    }
    {
      const double bb = acc[0], tol2 = uniform_double(d.rtol * d.rtol * bb);
      double rr = acc[1];
      if (rr > tol2 && bb != 0.0) {
        double rho = rr, rho_old = 1.0, alpha = 1.0, omega = 1.0;
        int it = 0;
        // shadow residual in registers (own rows).  BiCGStab accepts ANY fixed shadow vector with
        // (rh, r0) != 0; we take r0 rounded to fp32, which halves its register footprint.
        float2 rh[TROWS];
        {
          double a0[1] = {0.0};
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            rh[k] = make_float2(0.f, 0.f);
            if (row < n2) {
              const double2 r0 = Rl[row];
              rh[k] = make_float2((float)r0.x, (float)r0.y);
              a0[0] += (double)rh[k].x * r0.x + (double)rh[k].y * r0.y;
            }
          }
          block_sum1<1, TW / 64>(a0, red, rsel);
          rho = a0[0];  // (rh, r0)
        }
        AT_STAMP(4)
        while (it < d.maxit_u) {
          ++it;
          const double beta = (rho / rho_old) * (alpha / omega);
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            if (row < n2) {
              const double2 ri = Rl[row], pi = Pl[row];
              Pl[row] = make_double2(ri.x + beta * (pi.x - omega * vv[k].x), ri.y + beta * (pi.y - omega * vv[k].y));
              Yl[row] = make_double2(0.0, 0.0);
            }
          }
          __syncthreads();
          AT_STAMP(5)
          outflow_entries_add<TW>(v, bo_ent, -0.5 * mu, Pl, Yd);
          atomic_accumulate<(TPAIR > 1), TW, TPAIR>(v, Yd, tm, [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
            double2 xe[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xe[i] = Pl[E.dof[i]];
            elem_velocity(g, a, mu, xe, ye);
          });
          __syncthreads();
          AT_STAMP(6)
          double a1[1] = {0.0};
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            if (row < n2) {
              const double2 yv = Yl[row];
              vv[k] = make_double2(yv.x * f2d(idg[k].x), yv.y * f2d(idg[k].y));
              a1[0] += f2d(rh[k].x) * vv[k].x + f2d(rh[k].y) * vv[k].y;
            }
          }
          block_sum1<1, TW / 64>(a1, red, rsel);
          AT_STAMP(7)
          if (a1[0] == 0.0) break;
          alpha = rho / a1[0];
          // s = r - alpha v ; no early exit on |s| (saves a reduction; the check on |r| follows)
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            if (row < n2) {
              const double2 ri = Rl[row];
              Rl[row] = make_double2(ri.x - alpha * vv[k].x, ri.y - alpha * vv[k].y);
              Yl[row] = make_double2(0.0, 0.0);
            }
          }
          __syncthreads();  // publish s and the zeroed result vector
          AT_STAMP(8)
          outflow_entries_add<TW>(v, bo_ent, -0.5 * mu, Rl, Yd);
          atomic_accumulate<(TPAIR > 1), TW, TPAIR>(v, Yd, tm, [&](int e, const Geo& g, const ElemIdx& E, int ko, double2(&ye)[6]) {
            double2 xe[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xe[i] = Rl[E.dof[i]];
            elem_velocity(g, a, mu, xe, ye);
          });
          // x of the own rows: issue the global reads now, they are consumed after the reduction
          double2 xo[TROWS];
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            xo[k] = row < n2 ? xs[row] : make_double2(0.0, 0.0);
          }
          __syncthreads();
          AT_STAMP(9)
          double a3[2] = {0.0, 0.0};
          double2 t[TROWS];
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            t[k] = make_double2(0.0, 0.0);
            if (row < n2) {
              const double2 yv = Yl[row], sv = Rl[row];
              t[k] = make_double2(yv.x * f2d(idg[k].x), yv.y * f2d(idg[k].y));
              a3[0] += t[k].x * sv.x + t[k].y * sv.y;
              a3[1] += t[k].x * t[k].x + t[k].y * t[k].y;
            }
          }
          block_sum1<2, TW / 64>(a3, red, rsel);
          AT_STAMP(10)
          if (a3[1] == 0.0) break;
          omega = a3[0] / a3[1];
          double a4[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 0; k < TROWS; ++k) {
            const int row = tid + k * TW;
            if (row < n2) {
              const double2 pi = Pl[row], sv = Rl[row];
              xs[row] = make_double2(xo[k].x + alpha * pi.x + omega * sv.x, xo[k].y + alpha * pi.y + omega * sv.y);
              const double2 rn = make_double2(sv.x - omega * t[k].x, sv.y - omega * t[k].y);
              Rl[row] = rn;
              a4[0] += rn.x * rn.x + rn.y * rn.y;
              a4[1] += f2d(rh[k].x) * rn.x + f2d(rh[k].y) * rn.y;
            }
          }
          block_sum1<2, TW / 64>(a4, red, rsel);
          AT_STAMP(11)
          rr = a4[0];
          if (!(rr > tol2)) break;
          rho_old = rho;
          rho = a4[1];
          if (rho == 0.0 || omega == 0.0) break;
        }
        it_u += it;
      }
    }
    __syncthreads();  // xs (= u*) complete: the element loops of steps 2 and 3 gather it
    AT_STAMP(12)
  }
  if (tid == 0) {
    histc[0] = (double)(nhist < 5 ? nhist + 1 : 5);
    if (iters) iters[3 * b + 0] += it_u;
  }
}

// NTH = 1024 for the direct solver (its dense phases are latency-bound streams of 0.8 MB of factors: twice the loads
// in flight per CU); the CG variant keeps the 512-thread shape its reductions are written for.
template <bool K1_LDS, int NTH = WG>
__global__ __launch_bounds__(NTH) void at_pressure_kernel(mdq_ipcs_desc d, int32_t* iters) {
  extern __shared__ __align__(16) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);
  const double a = d.rho / d.dt, mu = d.mu;
  (void)a; (void)mu; (void)nv; (void)n2;
  double* red = smem;  // 64 doubles
  double* U = smem + 64;
  double* w = v.work;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);  // x of the velocity solve = u*
  double* pnew = reinterpret_cast<double*>(xs + 6 * (int64_t)d.N2);
  int rsel = 0;  // parity of the one-barrier reductions
  (void)rsel; (void)pnew; (void)red;
  double* px = U;
  double* pr = px + P.NVp;
  double* pp = pr + P.NVp;
  double* pq = pp + P.NVp;
  double* lK = pq + P.NVp;
  int32_t* lci = reinterpret_cast<int32_t*>(lK + d.NSE1);
  int32_t* lso = lci + d.NSE1 + (d.NSE1 & 1);
  double* escr1 = w;
  const int nsl1 = (nv + 63) >> 6;
  const int32_t* so1 = K1_LDS ? lso : v.sl1_off;
  const int32_t* ci1 = K1_LDS ? lci : v.sl1_col;
  const double* K1 = K1_LDS ? lK : v.K1s;
  int it_p = 0;
#ifdef MDQ_AT_TRACE
  const long long tq0_ = __builtin_amdgcn_s_memtime();
#endif
  {
    // ================= step 2: pressure
    if (K1_LDS && !d.pd_enabled) {
      const int ne1 = v.sl1_off[nsl1];
      for (int kk = tid; kk < ne1; kk += NTH) {
        lK[kk] = v.K1s[kk];
        lci[kk] = v.sl1_col[kk];
      }
      for (int kk = tid; kk <= nsl1; kk += NTH) lso[kk] = v.sl1_off[kk];
    }
    (void)escr1;
    for (int i = tid; i < nv; i += NTH) pr[i] = 0.0;
    __syncthreads();
    rhs2_accumulate_lds<NTH>(v, d, xs, v.p_n, pr);
    // own-row data of the scaling, loaded in front of the barrier that ends the accumulation
    constexpr int PRW = 2048 / NTH;   // own rows per thread held in registers (vertex capacity of the LDS modes < 2048)
    double sd_[PRW], pn_[PRW];
    unsigned fl_ = 0;
#pragma unroll
    for (int k = 0; k < PRW; ++k) {
      const int i = min(tid + k * NTH, nv - 1);
      sd_[k] = v.sdiagK[i];
      pn_[k] = v.p_n[i];
      fl_ |= v.bcp_flag[i] ? 1u << k : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PRW; ++k) {
      const int i = tid + k * NTH;
      if (i < nv) {
        pr[i] = ((fl_ >> k) & 1u) ? 0.0 : pr[i] / sd_[k];
        px[i] = pn_[k] * sd_[k];
      }
    }
    for (int i = tid + PRW * NTH; i < nv; i += NTH) {   // (nv beyond 2048: generic tail)
      const double sd = v.sdiagK[i];
      pr[i] = v.bcp_flag[i] ? 0.0 : pr[i] / sd;
      px[i] = v.p_n[i] * sd;
    }
    // (an environment whose factors could not be built on the device - mdq_ipcs_factorize_pressure, header nparts = 0 -
    // falls back to the Krylov solve)
    if (d.pd_enabled && d.pd_hdr[4 * (int64_t)b + 2] > 0) {
      const PdView pd = pd_view(d, b);
#ifdef MDQ_AT_TRACE
      { const long long tn_ = __builtin_amdgcn_s_memtime(); if (tid == 0 && b == 0) mdq_pt_trace_buf[0] += tn_ - tq0_; }
#endif
      pressure_direct<NTH>(pd, nv, pr, px, pp, pq, lK);
    } else {
      int itc = -1;
      // (the 1024-thread instance is the direct solver's shape - 128 VGPRs per thread: the register-hungry Krylov variants
      //  compiled into it cost 272 B of scratch per lane in round 3 without ever running there; an environment without
      //  factors takes the plain register CG below)
      if constexpr (NTH <= 512) {
        if (d.pcg_degree < 0)                    // two-level additive preconditioner (geometric aggregates)
          itc = cg_pressure_2l<NTH>(nv, so1, ci1, K1, v.coords, d.rtol, d.maxit_p, px, pr, pp, P.NVp, red, rsel);
        else if (nv <= 2 * NTH && d.pcg_degree > 0)   // rows in registers, Chebyshev-preconditioned (degree 1: Jacobi only)
          itc = cg_pressure_cheb<NTH>(nv, so1, ci1, K1, d.rtol, d.maxit_p, d.pcg_degree, px, pr, pp, red, rsel);
        if (itc < 0 && nv <= 2 * NTH && d.pcg_degree == 0 && MDQ_PCG_REGM)   // rows in registers, branch-free application
          itc = cg_pressure_regm<NTH>(nv, so1, ci1, K1, d.rtol, d.maxit_p, px, pr, pp, red, rsel);
      }
      if (itc >= 0)
        it_p += itc;
      else if (nv <= 2 * NTH)
        it_p += cg_pressure_reg<NTH>(nv, so1, ci1, K1, d.rtol, d.maxit_p, px, pr, pp, red, rsel);
      else
        it_p += cg_pressure(nv, so1, ci1, K1, d.rtol, d.maxit_p, px, pr, pp, pq, red);
    }
    for (int i = tid; i < nv; i += NTH) pnew[i] = px[i] / v.sdiagK[i];
    __syncthreads();

  }
  if (tid == 0 && iters) iters[3 * b + 1] += it_p;
}

#if MDQ_IN_PART(0)
static __global__ __launch_bounds__(WG) void at_correction_kernel(mdq_ipcs_desc d, int nsteps, int step, double* drag,
                                                            double* lift, int32_t* iters) {
  extern __shared__ __align__(16) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const EnvView v = env_view(d, b);
  const int n2 = v.n2, nv = v.nv;
  const LdsPlan P = lds_plan(d.N2, d.NV, d.NSE1);
  const double a = d.rho / d.dt, mu = d.mu;
  (void)a; (void)mu; (void)nv; (void)n2;
  double* red = smem;  // 64 doubles
  double* U = smem + 64;
  double* w = v.work;
  double2* xs = reinterpret_cast<double2*>(w + 12 * (int64_t)d.NT);  // x of the velocity solve = u*
  double* pnew = reinterpret_cast<double*>(xs + 6 * (int64_t)d.N2);
  int rsel = 0;  // parity of the one-barrier reductions
  (void)rsel; (void)pnew; (void)red;
  double2* Pl = reinterpret_cast<double2*>(U);
  double2* Yl = Pl + P.N2p;
  double* Yd = reinterpret_cast<double*>(Yl);
  AtMeta tm;
  at_prefetch(v, tm, 0);
  int it_m = 0;
  // ring of the last three corrections u_{n+1} - u* (in the slab region the assembled modes use for their history)
  double2* cring = reinterpret_cast<double2*>(w + work_hist_offset(d.NV, d.NT, d.NE));
  double* ccnt = reinterpret_cast<double*>(cring + 3 * (int64_t)d.N2);  // [0]: corrections stored, [1]: ring position
  const int nc = (int)ccnt[0], rp = (int)ccnt[1];
#ifdef MDQ_AT_TRACE
  long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  CT_STAMP(0)
  {
    // ================= step 3: velocity correction (mass solve, both components)
    double2 x[MF_ROWS], r[MF_ROWS], p[MF_ROWS];
    double ism[MF_ROWS];
    double am[2] = {0.0, 0.0};
    // Row loops as in at_velocity_kernel: clamped rows, all loads of a phase issued before their first use.
    int rc[MF_ROWS];
#pragma unroll
    for (int k = 0; k < MF_ROWS; ++k) rc[k] = min(tid + k * WG, n2 - 1);
    double* Dp = reinterpret_cast<double*>(Yl + P.N2p);   // [nv] pressure increment, staged for the element loop
    // initial guess of the correction: extrapolation in time of the stored ones (constant, linear, quadratic as the
    // ring fills): 5.0 CG iterations per step with the previous correction alone, 1.7 with the quadratic guess (the
    // velocity solve then needs 2.9 instead of 2.6 iterations: u_n now carries the full 1e-10 solver noise).
    // Dirichlet rows keep u* = g (their stored corrections are 0).
    auto load_delta = [&](double2(&dl)[MF_ROWS]) {
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) dl[k] = make_double2(0.0, 0.0);
      if (nc >= 1) {
        const double2* c1 = cring + (int64_t)((rp + 2) % 3) * d.N2;   // newest
        const double2* c2 = cring + (int64_t)((rp + 1) % 3) * d.N2;
        const double2* c3 = cring + (int64_t)rp * d.N2;               // oldest (overwritten at the end of this launch)
#pragma unroll
        for (int k = 0; k < MF_ROWS; ++k) dl[k] = c1[rc[k]];
        if (nc >= 2) {
          const double w1 = nc >= 3 ? 3.0 : 2.0, w2 = nc >= 3 ? -3.0 : -1.0;
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const double2 b2 = c2[rc[k]];
            dl[k] = make_double2(w1 * dl[k].x + w2 * b2.x, w1 * dl[k].y + w2 * b2.y);
          }
          if (nc >= 3) {
#pragma unroll
            for (int k = 0; k < MF_ROWS; ++k) {
              const double2 b3 = c3[rc[k]];
              dl[k] = make_double2(dl[k].x + b3.x, dl[k].y + b3.y);
            }
          }
        }
      }
    };
    // FUSED start (15 of 16 steps once the ring is full): with x0 = u* + delta0 the initial residual is
    //   f3 - M x0 = (M u* - g) - M (u* + delta0) = -g - M delta0,
    // ONE element loop (the right-hand-side operator applied to -delta0) instead of two (f3, then M x0), and no
    // lifting vector (delta0 vanishes on the Dirichlet rows).  What it cannot give is |b| of the stopping test
    // rtol |b|, which needs f3 itself: the value of the last exact start is used (it changes by < 1e-3 per step);
    // every 16th step, and while the ring fills, the exact two-loop start runs and refreshes it.
    const double bb_lag = ccnt[2];
    const int cstep = (int)ccnt[3];
    const bool fused = nc >= 3 && bb_lag > 0.0 && (cstep & 15) != 0;
    {
      // operator input (u*, or -delta0) and p_new - p_n staged in LDS: the element loop gathers from LDS, not L2
      double2 us[MF_ROWS];
      if (!fused) {
#pragma unroll
        for (int k = 0; k < MF_ROWS; ++k) us[k] = xs[rc[k]];
      } else {
        load_delta(us);
#pragma unroll
        for (int k = 0; k < MF_ROWS; ++k) us[k] = make_double2(-us[k].x, -us[k].y);
      }
      for (int i = tid; i < nv; i += WG) Dp[i] = pnew[i] - v.p_n[i];
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        if (row < n2) {
          Pl[row] = us[k];
          Yl[row] = make_double2(0.0, 0.0);
        }
      }
    }
    __syncthreads();
    atomic_accumulate<true>(v, Yd, tm, [&](int, const Geo& g, const ElemIdx& E, int, double2(&ye)[6]) {
      double2 ue[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ue[i] = Pl[E.dof[i]];
      double dp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) dp[i] = Dp[E.dof[i]];
      elem_rhs3(g, d.dt, ue, dp, ye);
    });
    // own rows of the global vectors of the next phase: issued in front of the barrier that ends the element loop
    double2 l3[MF_ROWS], dlt[MF_ROWS];
    unsigned flm = 0;
    if (!fused) load_delta(dlt);   // (the fused start reads delta0 back from its LDS copy below)
#pragma unroll
    for (int k = 0; k < MF_ROWS; ++k) {
      l3[k] = fused ? xs[rc[k]] : v.lift3[rc[k]];     // fused start: u* of the own rows (the LDS copy holds -delta0)
      ism[k] = v.sdiagM[rc[k]];
      flm |= v.bcu_flag[rc[k]] ? 1u << k : 0u;
    }
    __syncthreads();
    CT_STAMP(1)
    double bb;
    if (fused) {
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        r[k] = Yl[rc[k]];
        const double2 md = Pl[rc[k]];
        dlt[k] = make_double2(-md.x, -md.y);
      }
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        const bool fl = (flm >> k) & 1u;
        ism[k] = (row < n2 && !fl) ? 1.0 / ism[k] : 0.0;
        r[k] = make_double2(r[k].x * ism[k], r[k].y * ism[k]);          // 0 on constrained rows and past n2
        p[k] = r[k];
        am[1] += r[k].x * r[k].x + r[k].y * r[k].y;
        // scaled unknown S x0 on the free rows, u* = g on the Dirichlet rows
        x[k] = fl ? l3[k] : make_double2((l3[k].x + dlt[k].x) / (ism[k] != 0.0 ? ism[k] : 1.0),
                                          (l3[k].y + dlt[k].y) / (ism[k] != 0.0 ? ism[k] : 1.0));
        if (row >= n2) x[k] = make_double2(0.0, 0.0);
      }
      double a1r[1] = {am[1]};
      block_sum<1>(a1r, red);
      am[1] = a1r[0];
      bb = bb_lag;
      CT_STAMP(2)
      CT_STAMP(3)
    } else {
      double2 f3[MF_ROWS];
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        f3[k] = Yl[rc[k]];
        x[k] = Pl[rc[k]];      // u* (own row)
      }
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        const bool fl = (flm >> k) & 1u;
        ism[k] = (row < n2 && !fl) ? 1.0 / ism[k] : 0.0;
        if (!fl) x[k] = make_double2(x[k].x + dlt[k].x, x[k].y + dlt[k].y);
        if (row < n2) {
          Yl[row] = make_double2(0.0, 0.0);
          Pl[row] = x[k];      // stage S^-1 (S x0) = x0
          const double2 bi = fl ? x[k] : make_double2((f3[k].x - l3[k].x) * ism[k], (f3[k].y - l3[k].y) * ism[k]);
          am[0] += bi.x * bi.x + bi.y * bi.y;
        } else {
          f3[k] = make_double2(0.0, 0.0);
        }
      }
      // (x0 is not carried in registers across the element loop below - 28 VGPRs the loop's two interleaved triangles need,
      //  spilled to scratch in round 3: its own rows are read back from the staged copy, which the loop only reads)
      __syncthreads();
      CT_STAMP(2)
      // (one triangle at a time: beside f3 / ism / the prefetched metadata the two interleaved triangles of the other
      //  applications did not fit - 180 B of scratch per lane in round 3; this application runs once per launch)
      atomic_accumulate<MDQ_CORR_MX0_INTERLEAVE>(v, Yd, tm, [&](int, const Geo& g, const ElemIdx& E, int, double2(&ye)[6]) {
        double2 xe[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) xe[i] = Pl[E.dof[i]];
        elem_mass(g, xe, ye);
      });
      __syncthreads();
      CT_STAMP(3)
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        r[k] = p[k] = x[k] = make_double2(0.0, 0.0);
        if (row < n2) {
          const double2 ax = Yl[row];
          x[k] = Pl[row];
          r[k] = make_double2((f3[k].x - ax.x) * ism[k], (f3[k].y - ax.y) * ism[k]);  // 0 on constrained rows
          p[k] = r[k];
          am[1] += r[k].x * r[k].x + r[k].y * r[k].y;
          if (ism[k] != 0.0) x[k] = make_double2(x[k].x / ism[k], x[k].y / ism[k]);  // scaled unknown S x
        }
      }
      block_sum<2>(am, red);
      bb = am[0];
      if (tid == 0) ccnt[2] = bb;
    }
    CT_STAMP(4)
    {
      const double tol2 = d.rtol * d.rtol * bb;
      double rr = am[1];
      if (rr > tol2 && bb != 0.0) {
        int it = 0;
        while (it < d.maxit_m) {
          ++it;
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = tid + k * WG;
            if (row < n2) {
              Pl[row] = make_double2(p[k].x * ism[k], p[k].y * ism[k]);
              Yl[row] = make_double2(0.0, 0.0);
            }
          }
          __syncthreads();
          CT_STAMP(5)
          atomic_accumulate<true>(v, Yd, tm, [&](int, const Geo& g, const ElemIdx& E, int, double2(&ye)[6]) {
            double2 xe[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xe[i] = Pl[E.dof[i]];
            elem_mass(g, xe, ye);
          });
          __syncthreads();
          CT_STAMP(6)
          double a1[1] = {0.0};
          double2 q[MF_ROWS];
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            const int row = tid + k * WG;
            q[k] = make_double2(0.0, 0.0);
            if (row < n2) {
              const double2 yv = Yl[row];
              q[k] = make_double2(yv.x * ism[k], yv.y * ism[k]);
              a1[0] += p[k].x * q[k].x + p[k].y * q[k].y;
            }
          }
          block_sum1<1>(a1, red, rsel);
          CT_STAMP(7)
          if (!(a1[0] > 0.0)) break;
          const double alpha = rr / a1[0];
          double a2[1] = {0.0};
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k) {
            x[k] = make_double2(x[k].x + alpha * p[k].x, x[k].y + alpha * p[k].y);
            r[k] = make_double2(r[k].x - alpha * q[k].x, r[k].y - alpha * q[k].y);
            a2[0] += r[k].x * r[k].x + r[k].y * r[k].y;
          }
          block_sum1<1>(a2, red, rsel);
          CT_STAMP(8)
          const double rr_new = a2[0];
          if (!(rr_new > tol2)) break;
          const double beta = rr_new / rr;
          rr = rr_new;
#pragma unroll
          for (int k = 0; k < MF_ROWS; ++k)
            p[k] = make_double2(r[k].x + beta * p[k].x, r[k].y + beta * p[k].y);
        }
        it_m += it;
      }
    }

    // ================= update state + probes
    {
      double2 us[MF_ROWS];
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) us[k] = xs[rc[k]];
      double2* cnew = cring + (int64_t)rp * d.N2;
#pragma unroll
      for (int k = 0; k < MF_ROWS; ++k) {
        const int row = tid + k * WG;
        if (row < n2) {
          const double2 un = (ism[k] != 0.0) ? make_double2(x[k].x * ism[k], x[k].y * ism[k]) : x[k];
          v.u_n[row] = un;
          cnew[row] = (ism[k] != 0.0) ? make_double2(un.x - us[k].x, un.y - us[k].y) : make_double2(0.0, 0.0);
        }
      }
    }
    if (tid == 0) {
      ccnt[0] = (double)(nc < 3 ? nc + 1 : 3);
      ccnt[1] = (double)((rp + 1) % 3);
      ccnt[3] = (double)((cstep + 1) & 1023);
    }
    for (int i = tid; i < nv; i += WG) v.p_n[i] = pnew[i];
    __syncthreads();
    CT_STAMP(9)
    double dr, li;
    forces(v, d.mu, v.u_n, v.p_n, red, dr, li);
    CT_STAMP(10)
    if (tid == 0) {
      drag[(int64_t)b * nsteps + step] = dr;
      lift[(int64_t)b * nsteps + step] = li;
    }
  }
  if (tid == 0 && iters) iters[3 * b + 2] += it_m;
}
#endif


// ================================================================== host side

static void build_tables(RefTab& T) {
  const double s15 = std::sqrt(15.0);
  const double a1 = (6.0 - s15) / 21.0, a2 = (6.0 + s15) / 21.0;
  const double w0 = 9.0 / 40.0, w1 = (155.0 - s15) / 1200.0, w2 = (155.0 + s15) / 1200.0;
  const double pts[NQ][2] = {{1.0 / 3.0, 1.0 / 3.0}, {a1, a1}, {1.0 - 2.0 * a1, a1}, {a1, 1.0 - 2.0 * a1},
                             {a2, a2},               {1.0 - 2.0 * a2, a2}, {a2, 1.0 - 2.0 * a2}};
  const double wts[NQ] = {w0, w1, w1, w1, w2, w2, w2};
  auto basis = [](double xi, double eta, double* phi, double (*dphi)[2], double* psi) {
    const double lam[3] = {1.0 - xi - eta, xi, eta};
    const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
    const int EA[3] = {1, 0, 0}, EB[3] = {2, 2, 1};
    for (int k = 0; k < 3; ++k) {
      psi[k] = lam[k];
      phi[k] = lam[k] * (2.0 * lam[k] - 1.0);
      dphi[k][0] = (4.0 * lam[k] - 1.0) * dl[k][0];
      dphi[k][1] = (4.0 * lam[k] - 1.0) * dl[k][1];
      phi[3 + k] = 4.0 * lam[EA[k]] * lam[EB[k]];
      dphi[3 + k][0] = 4.0 * (lam[EA[k]] * dl[EB[k]][0] + lam[EB[k]] * dl[EA[k]][0]);
      dphi[3 + k][1] = 4.0 * (lam[EA[k]] * dl[EB[k]][1] + lam[EB[k]] * dl[EA[k]][1]);
    }
  };
  std::memset(&T, 0, sizeof(T));
  for (int q = 0; q < NQ; ++q) {
    T.qw[q] = 0.5 * wts[q];
    basis(pts[q][0], pts[q][1], T.qphi[q], T.qdphi[q], T.qpsi[q]);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        T.Mhat[i][j] += T.qw[q] * T.qphi[q][i] * T.qphi[q][j];
        for (int c = 0; c < 2; ++c)
          for (int dd = 0; dd < 2; ++dd) T.Ghat[c][dd][i][j] += T.qw[q] * T.qdphi[q][i][c] * T.qdphi[q][j][dd];
      }
  }
  const double g = 0.5 / std::sqrt(3.0);
  T.gx[0] = 0.5 - g;
  T.gx[1] = 0.5 + g;
  T.gw[0] = T.gw[1] = 0.5;
}

// the reference-element tables of THIS translation unit's kernels (c_tab has internal linkage)
static hipError_t upload_tables() {
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] {
    RefTab T;
    build_tables(T);
    err = hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &T, sizeof(T));
  });
  return err;
}

// launchers of the operator modes whose kernels live in the other parts of this file (each uploads its own tables)
struct EvolveArgs {
  const mdq_ipcs_desc* d;
  size_t lds;
  int nsteps;
  double *drag, *lift;
  int32_t* iters;
  hipStream_t st;
};
hipError_t part_launch_assembled(int mode, bool k1_lds, bool pg, const EvolveArgs& a);   // modes 0 / 1 (part 1)
hipError_t part_launch_tiles(int mode, bool k1_lds, bool pg, const EvolveArgs& a);       // modes 5 / 4 (part 2)
hipError_t part_launch_mf(bool k1_lds, const EvolveArgs& a);                              // mode 2     (part 3)

#if MDQ_IN_PART(1)
hipError_t part_launch_assembled(int mode, bool k1_lds, bool pg, const EvolveArgs& a) {
  if (hipError_t e = upload_tables(); e != hipSuccess) return e;
  if (mode == 1)
    return k1_lds ? launch_evolve<1, true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                  : launch_evolve<1, false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st);
  return pg ? launch_evolve<0, false, true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
            : (k1_lds ? launch_evolve<0, true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                      : launch_evolve<0, false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st));
}
#endif
#if MDQ_IN_PART(2)
hipError_t part_launch_tiles(int mode, bool k1_lds, bool pg, const EvolveArgs& a) {
  if (hipError_t e = upload_tables(); e != hipSuccess) return e;
  if (mode == 4)
    return k1_lds ? launch_evolve_team<true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                  : launch_evolve_team<false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st);
  if (mode == 7)
    return k1_lds ? launch_evolve_team_tiles<true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                  : launch_evolve_team_tiles<false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st);
  return pg ? launch_evolve<5, false, true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
            : (k1_lds ? launch_evolve<5, true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                      : launch_evolve<5, false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st));
}
#endif
#if MDQ_IN_PART(3)
hipError_t part_launch_mf(bool k1_lds, const EvolveArgs& a) {
  if (hipError_t e = upload_tables(); e != hipSuccess) return e;
  return k1_lds ? launch_evolve_mf<true>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st)
                : launch_evolve_mf<false>(a.d, a.lds, a.nsteps, a.drag, a.lift, a.iters, a.st);
}
#endif

#if MDQ_IN_PART(0)
static thread_local std::string g_err;

static int fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1;
}
static int fail_msg(const std::string& m) {
  g_err = m;
  return -2;
}

static int ensure_tables() {
  const hipError_t err = upload_tables();
  if (err != hipSuccess) return fail("hipMemcpyToSymbol(c_tab)", err);
  return 0;
}

static int check_desc(const mdq_ipcs_desc* d) {
  if (!d) return fail_msg("null descriptor");
  if (d->B <= 0 || d->NV <= 0 || d->NT <= 0 || d->NE <= 0) return fail_msg("bad batch sizes");
  if (d->N2 != d->NV + d->NE) return fail_msg("N2 must equal NV+NE");
  if (d->work_doubles < (int64_t)d->B * work_per_env(d->NV, d->NT, d->NE))
    return fail_msg("workspace too small (see mdq_ipcs_workspace_doubles)");
  if (!(d->dt > 0.0) || !(d->rho > 0.0)) return fail_msg("dt and rho must be positive");
  return 0;
}

#endif  // part 0
}  // namespace mdq

#if MDQ_IN_PART(0)
// shared by the other translation units of the library (mdq_internal.h)
int mdq_set_error(const char* msg) {
  mdq::g_err = msg;
  return -2;
}

using namespace mdq;

extern "C" {

int mdq_abi_version(void) { return MDQ_ABI_VERSION; }

const char* mdq_last_error(void) { return g_err.c_str(); }

int64_t mdq_ipcs_workspace_doubles(int32_t B, int32_t NV, int32_t NT, int32_t NE) {
  return (int64_t)B * work_per_env(NV, NT, NE);
}

int mdq_ipcs_reset_history(const mdq_ipcs_desc* d, int32_t* iters, void* stream) {
  if (int rc = check_desc(d)) return rc;
  hipLaunchKernelGGL(reset_history_kernel, dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *d, iters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("reset_history_kernel launch", e);
  return 0;
}

int mdq_ipcs_assemble(const mdq_ipcs_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (int rc = ensure_tables()) return rc;
  hipLaunchKernelGGL(assemble_kernel, dim3(d->B), dim3(WG), 0, (hipStream_t)stream, *d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("assemble_kernel launch", e);
  return 0;
}

int mdq_ipcs_setup_matfree(const mdq_ipcs_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (!d->coords || !d->cell_dofs || !d->cell_outflow || !d->g1_ptr || !d->g1_src || !d->g2_ptr || !d->g2_src ||
      !d->sl1_off || !d->sl1_col || !d->bcu_flag || !d->bcu_gx || !d->bcp_flag || !d->geom || !d->lift1 || !d->lift3 ||
      !d->idiag1 || !d->sdiagM || !d->sdiagK || !d->K1s)
    return fail_msg("mdq_ipcs_setup_matfree: incomplete descriptor");
  if (d->nbo && (!d->bo_rows || !d->bo_ptr || !d->bo_col || !d->bo_src || !d->bo_val))
    return fail_msg("mdq_ipcs_setup_matfree: outflow row list incomplete");
  if (int rc = ensure_tables()) return rc;
  // staged instance when the mesh's per-cell / per-dof data fit the LDS beside the kernel's static tables
  const size_t stage_bytes = 52 * (size_t)d->NT + 9 * (size_t)d->N2 + 9 * (size_t)d->NV + 64;
  hipError_t e;
  // + the slot lists (16-bit: slot ids < 6 NT <= 65535, pointers likewise) when they fit as well
  const size_t list_bytes = 2 * ((size_t)d->N2 + 1 + 6 * (size_t)d->NT + (size_t)d->NV + 1 + 3 * (size_t)d->NT) + 2;
  if (stage_bytes + list_bytes <= 156 * 1024 && 6 * (size_t)d->NT <= 65535) {
    const size_t bytes = stage_bytes + list_bytes;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&setup_matfree_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail("setup_matfree_kernel attribute", e);
    hipLaunchKernelGGL((setup_matfree_kernel<true, true>), dim3(d->B), dim3(SWG), bytes, (hipStream_t)stream, *d);
  } else if (stage_bytes <= 156 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&setup_matfree_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)stage_bytes);
    if (e != hipSuccess) return fail("setup_matfree_kernel attribute", e);
    hipLaunchKernelGGL(setup_matfree_kernel<true>, dim3(d->B), dim3(SWG), stage_bytes, (hipStream_t)stream, *d);
  } else if (const size_t idx_bytes = 12 * (size_t)d->NT + (size_t)d->N2 + 9 * (size_t)d->NV + 64;
             idx_bytes <= 156 * 1024 && d->N2 <= 65535 && std::getenv("MDQ_SETUP_UNSTAGED") == nullptr) {
    // (the index data alone: cell dofs, flags, the P1 scaling; MDQ_SETUP_UNSTAGED=1: the unstaged instance, A / B switch)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&setup_matfree_kernel<true, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)idx_bytes);
    if (e != hipSuccess) return fail("setup_matfree_kernel attribute", e);
    hipLaunchKernelGGL((setup_matfree_kernel<true, false, false>), dim3(d->B), dim3(SWG), idx_bytes, (hipStream_t)stream, *d);
  } else {
    hipLaunchKernelGGL(setup_matfree_kernel<false>, dim3(d->B), dim3(SWG), 0, (hipStream_t)stream, *d);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail("setup_matfree_kernel launch", e);
  return 0;
}

static int ipcs_evolve_impl(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift, int32_t* iters,
                            void* stream, double* kernel_ms);

int mdq_ipcs_evolve(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift, int32_t* iters,
                    void* stream) {
  return ipcs_evolve_impl(d, nsteps, drag, lift, iters, stream, nullptr);
}

int mdq_ipcs_evolve_timed(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift, int32_t* iters,
                          void* stream, double* kernel_ms) {
  if (!kernel_ms) return fail_msg("kernel_ms is required");
  if (d && d->mode != 3 && d->mode != -1) return fail_msg("per-kernel timing exists for the three-kernel mode 3 only");
  return ipcs_evolve_impl(d, nsteps, drag, lift, iters, stream, kernel_ms);
}

static int ipcs_evolve_impl(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift, int32_t* iters,
                            void* stream, double* kernel_ms) {
  if (int rc = check_desc(d)) return rc;
  if (nsteps <= 0) return fail_msg("nsteps must be positive");
  if (!drag || !lift) return fail_msg("drag/lift output pointers are required");
  if (int rc = ensure_tables()) return rc;
  const LdsPlan P = lds_plan(d->N2, d->NV, d->NSE1);
  const size_t LDS_MAX = 160 * 1024, red_bytes = 64 * sizeof(double);
  // pressure CG vectors beyond the LDS: kept in the workspace slab (modes 0 / 5; Jacobi-CG only - the factorisation has
  // its own, smaller limits and reports such a mesh through pd_status)
  const bool pg = red_bytes + P.prs_vec_bytes > LDS_MAX;
  // (pd_enabled with pg: the factors' own limits are far below this size - every such environment has nparts = 0 and
  //  takes the Krylov branch anyway; a small mesh riding along in the big layout is solved by CG as well)
  const bool k1_lds = !pg && !d->pd_enabled && red_bytes + P.prs_vec_bytes + P.prs_mat_bytes <= LDS_MAX;
  int mode = d->mode;
  // mode 5: the element tile (+ the stage of a chunk's input rows behind it when the packed local maps are given and fit)
  const size_t tile_bytes = tile_lds_bytes(*d);
  if (mode == 6) return fail_msg("unknown operator mode 6");
  if (mode < 0 || mode > 7) {  // auto: fastest variant that fits
    mode = 0;
    if (red_bytes + P.vel1_bytes <= LDS_MAX) mode = 1;
    if (red_bytes + P.vel2_bytes <= LDS_MAX && d->N2 <= MF_ROWS * WG) mode = 2;
    // (mode -2 = auto among the BITWISE REPRODUCIBLE variants: everything but the LDS-atomic mode 3)
    if (d->mode != -2 && red_bytes + P.vel3_bytes <= LDS_MAX && d->N2 <= MF_ROWS * WG) mode = 3;
    // a mesh beyond the LDS-resident vectors: the element tiles with global vectors (mode 5: ~5x less memory traffic than
    // the assembled operators and no sparsity pattern), when the tile maps were handed over
    if (mode == 0 && d->cell_outflow && ((d->mf_scat && d->mf_tptr) || (d->g2_ptr && d->g2_src)) && std::getenv("MDQ_NO_MODE5") == nullptr) mode = 5;
    if (mode == 0) {
      // a mesh that only fits the assembled global-memory path: two workgroups per environment while the batch leaves
      // at least half of the chip idle even so (measured on ys930 red-refined, ms per step one / two workgroups: B = 1
      // 13.7 / 8.0, 32: 16.7 / 11.9, 64: 24.5 / 22.0, 96: 26.0 / 29.3, 128: 20.0 / 26.1 - from ~64 environments on the
      // step is bound by the memory system as a whole, and the team barriers' cache write-backs / invalidations only add)
      const int ncu = team_cus((hipStream_t)stream);
      if (!pg && 2 * TEAM * d->B <= ncu) mode = 4;
    }
    // the element tiles with TWO workgroups per environment while the batch leaves half of the chip idle (BASELINE batch of
    // 128 refined meshes: 11.3 -> ms per step; MDQ_NO_TEAM_TILES=1 is the A / B switch)
    if (mode == 5 && !pg && std::getenv("MDQ_NO_TEAM_TILES") == nullptr) {
      if (TEAM * d->B <= team_cus((hipStream_t)stream)) mode = 7;
    }
  }
  if (kernel_ms && mode != 3) return fail_msg("per-kernel timing exists for the three-kernel mode 3 only");
  if (mode == 3 && (red_bytes + P.vel3_bytes > LDS_MAX || d->N2 > MF_ROWS * WG))
    return fail_msg("mode 3 needs N2 <= 3584 and three velocity vectors in LDS");
  if (mode == 2 && (red_bytes + P.vel2_bytes > LDS_MAX || d->N2 > MF_ROWS * WG))
    return fail_msg("matrix-free tile mode needs N2 <= 3584 and the x stage + element tile in LDS");
  if (mode == 1 && red_bytes + P.vel1_bytes > LDS_MAX) return fail_msg("LDS gather vectors do not fit");
  if ((mode == 5 || mode == 7) && (!d->cell_outflow || !((d->mf_scat && d->mf_tptr) || (d->g2_ptr && d->g2_src))))
    return fail_msg("modes 5 / 7 need cell_outflow and the tile maps (mf_scat, mf_tptr) or the dof <- slot lists (g2_ptr, g2_src)");
  if (pg && mode != 0 && mode != 5)
    return fail_msg("mesh too large for the LDS-resident pressure vectors of this operator mode (use mode -1, 0 or 5)");
  // (K1 values alias the scratch vector: CG does not use it; pg: the tables of the two-level preconditioner)
  size_t u = pg ? tl_extra_bytes(d->NV) + sizeof(double) * NAG * NAG : P.prs_vec_bytes + (k1_lds ? P.prs_mat_bytes : 0);
  const size_t vel = mode == 3 ? P.vel3_bytes : (mode == 2 ? P.vel2_bytes : (mode == 1 ? P.vel1_bytes : ((mode == 5 || mode == 7) ? tile_bytes : 0)));
  if (vel > u) u = vel;
  const size_t lds = red_bytes + u;
  hipError_t e;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 3) {
    const size_t lds_v = red_bytes + P.vel3_bytes;
    const size_t lds_p = red_bytes + P.prs_vec_bytes + (k1_lds ? P.prs_mat_bytes : 0);
    const size_t lds_c = red_bytes + 2 * sizeof(double2) * (size_t)P.N2p + sizeof(double) * (size_t)P.NVp;  // p, result, dp
    // once per process (thread-safe: several env groups call this entry point concurrently)
    static const hipError_t attr_err = [] {
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_velocity_kernel<WG, MF_ROWS, AT_PAIR>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ == hipSuccess)
        e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_velocity_kernel<768, 5, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ == hipSuccess)
        e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_pressure_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ == hipSuccess)
        e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_pressure_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ == hipSuccess)
        e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_pressure_kernel<false, 1024>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ == hipSuccess)
        e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&at_correction_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      return e_;
    }();
    if (attr_err != hipSuccess) return fail("hipFuncSetAttribute(mode 3 kernels)", attr_err);
    // velocity kernel variant: 12 waves x 1 triangle stream (3 waves per SIMD, <= 170 VGPRs) or 8 waves x 2 interleaved
    // triangle streams (2 per SIMD, 256 VGPRs); MDQ_AT_WG=512|768 overrides
    static const int vel_wg = [] {
      const char* s_ = std::getenv("MDQ_AT_WG");
      const int w_ = s_ ? std::atoi(s_) : 768;
      return w_ == 512 ? 512 : 768;
    }();
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (kernel_ms) {
      for (int i = 0; i < 4; ++i)
        if ((e = hipEventCreate(&ev[i])) != hipSuccess) return fail("hipEventCreate", e);
      kernel_ms[0] = kernel_ms[1] = kernel_ms[2] = 0.0;
    }
    for (int step = 0; step < nsteps; ++step) {
      if (kernel_ms) hipEventRecord(ev[0], st);
      if (vel_wg == 768)
        hipLaunchKernelGGL((at_velocity_kernel<768, 5, 1>), dim3(d->B), dim3(768), lds_v, st, *d, iters);
      else
        hipLaunchKernelGGL((at_velocity_kernel<WG, MF_ROWS, AT_PAIR>), dim3(d->B), dim3(WG), lds_v, st, *d, iters);
      if (kernel_ms) hipEventRecord(ev[1], st);
      if (d->pd_enabled)
        hipLaunchKernelGGL((at_pressure_kernel<false, 1024>), dim3(d->B), dim3(1024), lds_p, st, *d, iters);
      else if (k1_lds)
        hipLaunchKernelGGL(at_pressure_kernel<true>, dim3(d->B), dim3(WG), lds_p, st, *d, iters);
      else
        hipLaunchKernelGGL(at_pressure_kernel<false>, dim3(d->B), dim3(WG), lds_p, st, *d, iters);
      if (kernel_ms) hipEventRecord(ev[2], st);
      hipLaunchKernelGGL(at_correction_kernel, dim3(d->B), dim3(WG), lds_c, st, *d, nsteps, step, drag, lift, iters);
      if (kernel_ms) {
        hipEventRecord(ev[3], st);
        if ((e = hipEventSynchronize(ev[3])) != hipSuccess) return fail("hipEventSynchronize", e);
        for (int i = 0; i < 3; ++i) {
          float ms = 0.f;
          hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
          kernel_ms[i] += ms;
        }
      }
    }
    if (kernel_ms)
      for (int i = 0; i < 4; ++i) hipEventDestroy(ev[i]);
    e = hipGetLastError();
  } else {
    const EvolveArgs a{d, lds, nsteps, drag, lift, iters, st};
    if (mode == 2)
      e = part_launch_mf(k1_lds, a);
    else if (mode == 4 || mode == 5 || mode == 7)
      e = part_launch_tiles(mode, k1_lds, pg, a);
    else
      e = part_launch_assembled(mode, k1_lds, pg, a);
  }
  if (e != hipSuccess) return fail("evolve_kernel launch", e);
  return 0;
}

int mdq_probe_forces(const mdq_ipcs_desc* d, int32_t nfields, const double* u, const double* p, double* drag,
                     double* lift, void* stream) {
  // needs only: B, NV, NT, NE, N2, NAF, mu, nv, nt, ne, naf, coords, cell_dofs, af_facets
  if (!d || d->B <= 0 || d->N2 != d->NV + d->NE || !d->coords || !d->cell_dofs || !d->af_facets)
    return fail_msg("mdq_probe_forces: incomplete mesh descriptor");
  if (nfields <= 0 || !u || !p || !drag || !lift) return fail_msg("bad probe arguments");
  if (int rc = ensure_tables()) return rc;
  hipLaunchKernelGGL(probe_kernel, dim3(d->B), dim3(WG), 0, (hipStream_t)stream, *d, nfields, u, p, drag, lift);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("probe_kernel launch", e);
  return 0;
}

// DOLFIN MeshSmoothing::smooth restated for the host (flow_solver.py:65-67, 236-237).
int mdq_smooth_host(double* x, int32_t nv, const int32_t* cells, int32_t nt, const int64_t* nbr_ptr,
                    const int64_t* nbr, const int64_t* vc_ptr, const int64_t* vc, const uint8_t* on_boundary,
                    int32_t iterations) {
  if (!x || !cells || !nbr_ptr || !nbr || !vc_ptr || !vc || !on_boundary) return fail_msg("null argument");
  (void)nt;
  const double DOLFIN_EPS = 3.0e-16;
  for (int it = 0; it < iterations; ++it) {
    for (int v = 0; v < nv; ++v) {
      if (on_boundary[v]) continue;
      const double px = x[2 * v], py = x[2 * v + 1];
      double cx = 0.0, cy = 0.0;
      const int64_t n0 = nbr_ptr[v], n1 = nbr_ptr[v + 1];
      for (int64_t k = n0; k < n1; ++k) {
        cx += x[2 * nbr[k]];
        cy += x[2 * nbr[k] + 1];
      }
      const double cnt = (double)(n1 - n0);
      cx /= cnt;
      cy /= cnt;
      double rmin = 0.0;
      for (int64_t k = vc_ptr[v]; k < vc_ptr[v + 1]; ++k) {
        const int64_t c = vc[k] / 3, loc = vc[k] % 3;
        const int32_t o0 = cells[3 * c + (loc == 0 ? 1 : 0)], o1 = cells[3 * c + (loc == 2 ? 1 : 2)];
        const double ax = x[2 * o0], ay = x[2 * o0 + 1];
        const double tx = x[2 * o1] - ax, ty = x[2 * o1 + 1] - ay;
        double nx = ty, ny = -tx;
        const double nn = std::sqrt(nx * nx + ny * ny);
        nx /= nn;
        ny /= nn;
        const double r = std::fabs(nx * (px - ax) + ny * (py - ay));
        rmin = (rmin == 0.0) ? r : (r < rmin ? r : rmin);
      }
      const double dx = cx - px, dy = cy - py;
      const double r = std::sqrt(dx * dx + dy * dy);
      if (r < DOLFIN_EPS) continue;
      const double step = (0.5 * rmin < r) ? 0.5 * rmin : r;
      x[2 * v] = px + step * dx / r;
      x[2 * v + 1] = py + step * dy / r;
    }
  }
  return 0;
}

}  // extern "C"
#endif  // part 0
