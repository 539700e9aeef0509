// Shared device-side helpers for the gfx950 kernels (wave64, 1024-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdq {

constexpr int WG = 512;        // threads per workgroup = 8 wave64 (2 per SIMD, 256-VGPR budget); one workgroup per environment
constexpr int NWAVE = WG / 64;
constexpr int NQ = 7;          // Radon 7-point rule, exact to degree 5

// Reference-element tables (filled on the host once, see mdq_tables.cpp part of mdq_lib.hip)
struct RefTab {
  double qw[NQ];               // quadrature weights (sum = 1/2)
  double qphi[NQ][6];          // P2 basis values at the quadrature points
  double qdphi[NQ][6][2];      // P2 reference gradients
  double qpsi[NQ][3];          // P1 basis values
  double Mhat[6][6];           // int phi_i phi_j over the reference triangle
  double Ghat[2][2][6][6];     // int d_c phi_i d_d phi_j over the reference triangle
  double gx[2], gw[2];         // 2-point Gauss rule on [0,1]
};

// ---- wave64 sum with DPP (VALU cross-lane moves; __shfl_* lowers to ds_bpermute round trips
// through the LDS crossbar, ~10x slower for a 6-step fp64 butterfly) ---------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
  // lanes disabled by ROW_MASK and lanes shifted in from outside the row read 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, returned in every lane (fixed association order => deterministic)
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_get<0x111, 0xF>(v);  // row_shr:1
  v += dpp_get<0x112, 0xF>(v);  // row_shr:2
  v += dpp_get<0x114, 0xF>(v);  // row_shr:4
  v += dpp_get<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of each row of 16 holds the row sum
  v += dpp_get<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Deterministic workgroup-wide sum of N values; every thread returns the same bits.
// `lds` must hold NWAVE*N doubles.  Two barriers.
template <int N, int NW = NWAVE>
__device__ __forceinline__ void block_sum(double (&v)[N], double* lds) {
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = wave_sum(v[n]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < N; ++n) lds[w * N + n] = v[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < N; ++n) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += lds[i * N + n];
    v[n] = s;
  }
}

// local edge k (opposite local vertex k) has endpoints (EA[k], EB[k])
__device__ __forceinline__ int edge_a(int k) { return k == 0 ? 1 : 0; }
__device__ __forceinline__ int edge_b(int k) { return k == 2 ? 1 : 2; }

// P1 / P2 basis at a reference point
__device__ __forceinline__ void p1_eval(double xi, double eta, double (&psi)[3]) {
  psi[0] = 1.0 - xi - eta;
  psi[1] = xi;
  psi[2] = eta;
}

__device__ __forceinline__ void p2_eval(double xi, double eta, double (&phi)[6], double (&dphi)[6][2]) {
  const double lam[3] = {1.0 - xi - eta, xi, eta};
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    phi[k] = lam[k] * (2.0 * lam[k] - 1.0);
    dphi[k][0] = (4.0 * lam[k] - 1.0) * dl[k][0];
    dphi[k][1] = (4.0 * lam[k] - 1.0) * dl[k][1];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int a = edge_a(k), b = edge_b(k);
    phi[3 + k] = 4.0 * lam[a] * lam[b];
    dphi[3 + k][0] = 4.0 * (lam[a] * dl[b][0] + lam[b] * dl[a][0]);
    dphi[3 + k][1] = 4.0 * (lam[a] * dl[b][1] + lam[b] * dl[a][1]);
  }
}

// Geometry of local edge k of a cell with vertices X[3][2]: outward unit normal,
// length, and the reference coordinates of its end points.
struct Facet {
  double nx, ny, len;
  double ra[2], rb[2];
};

__device__ __forceinline__ double sel3(int k, double v0, double v1, double v2) {
  return k == 0 ? v0 : (k == 1 ? v1 : v2);
}

// (no runtime-indexed register arrays: they would be demoted to scratch memory)
__device__ __forceinline__ Facet facet_geometry(const double (&X)[3][2], int k) {
  // end points a<b of local edge k, opposite vertex k
  const double ax = sel3(k, X[1][0], X[0][0], X[0][0]), ay = sel3(k, X[1][1], X[0][1], X[0][1]);
  const double bx = sel3(k, X[2][0], X[2][0], X[1][0]), by = sel3(k, X[2][1], X[2][1], X[1][1]);
  const double kx = sel3(k, X[0][0], X[1][0], X[2][0]), ky = sel3(k, X[0][1], X[1][1], X[2][1]);
  Facet f;
  const double tx = bx - ax, ty = by - ay;
  f.len = sqrt(tx * tx + ty * ty);
  double nx = ty / f.len, ny = -tx / f.len;
  if (nx * (kx - ax) + ny * (ky - ay) > 0.0) {
    nx = -nx;
    ny = -ny;
  }
  f.nx = nx;
  f.ny = ny;
  // reference coordinates of local vertices: v0=(0,0), v1=(1,0), v2=(0,1)
  f.ra[0] = sel3(k, 1.0, 0.0, 0.0);
  f.ra[1] = 0.0;
  f.rb[0] = sel3(k, 0.0, 0.0, 1.0);
  f.rb[1] = sel3(k, 1.0, 1.0, 0.0);
  return f;
}

// One-barrier variant: `lds` holds two buffers of NWAVE*N doubles used alternately (`sel` flips on
// every call, workgroup-uniformly).  A buffer is rewritten two calls later, i.e. after the barrier
// of the call in between, which every thread passes only after it has finished reading.
template <int N, int NW = NWAVE>
__device__ __forceinline__ void block_sum1(double (&v)[N], double* lds, int& sel) {
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = wave_sum(v[n]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* buf = lds + sel * (NW * 2);
  sel ^= 1;
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < N; ++n) buf[w * N + n] = v[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < N; ++n) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += buf[i * N + n];
    v[n] = s;
  }
}

}  // namespace mdq
