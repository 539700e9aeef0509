// Per-triangle Taylor-Hood (P2/P1) element operators with compile-time reference data.
//
// All quadrature points / basis values are `constexpr`: after full unrolling every
// coefficient is an instruction literal (no table loads, no SGPR pressure) and zero
// coefficients disappear (`fma_c`).  Follows the UFL forms of flow_solver.py:98-120.
#pragma once
#include "mdq_device.h"

namespace mdq {

// ---- compile-time reference element ---------------------------------------
constexpr double c_lam(double x, double y, int k) { return k == 0 ? 1.0 - x - y : (k == 1 ? x : y); }
constexpr double c_dl(int k, int c) { return k == 0 ? -1.0 : ((k == 1) == (c == 0) ? 1.0 : 0.0); }
constexpr int c_ea(int k) { return k == 0 ? 1 : 0; }
constexpr int c_eb(int k) { return k == 2 ? 1 : 2; }
constexpr double c_phi(double x, double y, int i) {
  return i < 3 ? c_lam(x, y, i) * (2.0 * c_lam(x, y, i) - 1.0)
               : 4.0 * c_lam(x, y, c_ea(i - 3)) * c_lam(x, y, c_eb(i - 3));
}
constexpr double c_dphi(double x, double y, int i, int c) {
  return i < 3 ? (4.0 * c_lam(x, y, i) - 1.0) * c_dl(i, c)
               : 4.0 * (c_lam(x, y, c_ea(i - 3)) * c_dl(c_eb(i - 3), c) + c_lam(x, y, c_eb(i - 3)) * c_dl(c_ea(i - 3), c));
}
// 360 * int phi_i phi_j over the reference triangle
constexpr double c_m360(int i, int j) {
  return (i < 3 && j < 3) ? (i == j ? 6.0 : -1.0)
         : (i >= 3 && j >= 3) ? (i == j ? 32.0 : 16.0)
         : ((i < 3 ? j - 3 : i - 3) == (i < 3 ? i : j) ? -4.0 : 0.0);
}

// Radon 7-point rule (degree 5), weights sum to 1/2
constexpr double Q7A1 = 0.10128650732345633880098736191512383;
constexpr double Q7A2 = 0.47014206410511508977044120951344760;
constexpr double Q7W0 = 0.1125;
constexpr double Q7W1 = 0.06296959027241357629784197275009067;
constexpr double Q7W2 = 0.06619707639425309036882469391657600;
constexpr double q7x(int q) {
  return q == 0 ? 1.0 / 3.0 : q == 1 ? Q7A1 : q == 2 ? 1.0 - 2.0 * Q7A1 : q == 3 ? Q7A1 : q == 4 ? Q7A2 : q == 5 ? 1.0 - 2.0 * Q7A2 : Q7A2;
}
constexpr double q7y(int q) {
  return q == 0 ? 1.0 / 3.0 : q == 1 ? Q7A1 : q == 2 ? Q7A1 : q == 3 ? 1.0 - 2.0 * Q7A1 : q == 4 ? Q7A2 : q == 5 ? Q7A2 : 1.0 - 2.0 * Q7A2;
}
constexpr double q7w(int q) { return q == 0 ? Q7W0 : (q < 4 ? Q7W1 : Q7W2); }

// edge-midpoint rule (degree 2), weights 1/6: point q = midpoint of the edge opposite vertex q
constexpr double q3x(int q) { return q == 1 ? 0.0 : 0.5; }
constexpr double q3y(int q) { return q == 2 ? 0.0 : 0.5; }

// acc += coef * v, dropped at compile time when the (constant-folded) coefficient is zero
__device__ __forceinline__ void fma_c(double& acc, double coef, double v) {
  if (coef == 1.0)
    acc += v;
  else if (coef == -1.0)
    acc -= v;
  else if (coef != 0.0)
    acc += coef * v;
}

struct Geo {
  double j00, j01, j10, j11, det;  // Jinv[c][a] (reference row c, physical column a) and |det J|
};

// ---- operator applications (matrix-free) -------------------------------------

// y = M_e x  (P2 mass, both components)
__device__ __forceinline__ void elem_mass(const Geo& g, const double2 (&x)[6], double2 (&y)[6]) {
  const double s = g.det * (1.0 / 360.0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      fma_c(ax, c_m360(i, j), x[j].x);
      fma_c(ay, c_m360(i, j), x[j].y);
    }
    y[i] = make_double2(s * ax, s * ay);
  }
}

// y = (a M_e + mu Keps_e) x : `rho/dt*(u,v) + mu*inner(eps(u),eps(v))`, the volume part of lhs(F1)
// (flow_solver.py:106-112).  Stiffness by the edge-midpoint rule (exact), mass in closed form.
__device__ __forceinline__ void elem_velocity(const Geo& g, double a, double mu, const double2 (&x)[6],
                                              double2 (&y)[6]) {
  elem_mass(g, x, y);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    y[i].x *= a;
    y[i].y *= a;
  }
  const double w = mu * g.det * (1.0 / 6.0);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    // reference gradients of both components
    double ax0 = 0, ax1 = 0, ay0 = 0, ay1 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fma_c(ax0, c_dphi(q3x(q), q3y(q), i, 0), x[i].x);
      fma_c(ax1, c_dphi(q3x(q), q3y(q), i, 1), x[i].x);
      fma_c(ay0, c_dphi(q3x(q), q3y(q), i, 0), x[i].y);
      fma_c(ay1, c_dphi(q3x(q), q3y(q), i, 1), x[i].y);
    }
    // physical: d_a u = sum_c Jinv[c][a] dref_c
    const double uxx = g.j00 * ax0 + g.j10 * ax1, uxy = g.j01 * ax0 + g.j11 * ax1;  // d_x ux, d_y ux
    const double uyx = g.j00 * ay0 + g.j10 * ay1, uyy = g.j01 * ay0 + g.j11 * ay1;
    const double exy = 0.5 * (uxy + uyx);
    // flux F_ac = w * eps_ac ; test function v = phi_i e_c : sum_a F_ac d_a phi_i
    const double Fxx = w * uxx, Fxy = w * exy, Fyy = w * uyy;
    // back to reference directions: (Jinv F)_{c'} = sum_a Jinv[c'][a] F_a
    const double rx0 = g.j00 * Fxx + g.j01 * Fxy, rx1 = g.j10 * Fxx + g.j11 * Fxy;  // component x: (F_xx, F_yx)
    const double ry0 = g.j00 * Fxy + g.j01 * Fyy, ry1 = g.j10 * Fxy + g.j11 * Fyy;  // component y: (F_xy, F_yy)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fma_c(y[i].x, c_dphi(q3x(q), q3y(q), i, 0), rx0);
      fma_c(y[i].x, c_dphi(q3x(q), q3y(q), i, 1), rx1);
      fma_c(y[i].y, c_dphi(q3x(q), q3y(q), i, 0), ry0);
      fma_c(y[i].y, c_dphi(q3x(q), q3y(q), i, 1), ry1);
    }
  }
}

// y += coef * B_e x with B the outflow-facet term of F1:
// (B x)^c_i = int_facet phi_i n_d d_c x_d ds  (`dot(mu*nabla_grad(U)*n, v)*ds`, flow_solver.py:109)
__device__ inline void elem_outflow_add(const Geo& g, const double (&X)[3][2], int k, double coef,
                                        const double2 (&x)[6], double2 (&y)[6]) {
  const Facet f = facet_geometry(X, k);
  const double gs[2] = {0.5 - 0.28867513459481288225, 0.5 + 0.28867513459481288225};
  for (int q = 0; q < 2; ++q) {
    const double s = gs[q], w = coef * 0.5 * f.len;
    const double xi = f.ra[0] + s * (f.rb[0] - f.ra[0]);
    const double eta = f.ra[1] + s * (f.rb[1] - f.ra[1]);
    double phi[6], dphi[6][2];
    p2_eval(xi, eta, phi, dphi);
    double tx = 0.0, ty = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double gx = g.j00 * dphi[j][0] + g.j10 * dphi[j][1];
      const double gy = g.j01 * dphi[j][0] + g.j11 * dphi[j][1];
      const double un = x[j].x * f.nx + x[j].y * f.ny;
      tx += gx * un;
      ty += gy * un;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      y[i].x += w * phi[i] * tx;
      y[i].y += w * phi[i] * ty;
    }
  }
}

// ---- right-hand sides ------------------------------------------------------------

// step 1 (flow_solver.py:106-112): local vector of L1 = rhs(F1), volume part
//   r^c_i = int [ (a u_c - rho (u.grad)u_c) phi_i + (-mu eps(u)_{ac} + p delta_ac) d_a phi_i ]
__device__ __forceinline__ void elem_rhs1_vol(const Geo& g, double a, double mu, double rho, const double2 (&u)[6],
                                              const double (&p)[3], double2 (&r)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = make_double2(0.0, 0.0);
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    double ux = 0, uy = 0, ax0 = 0, ax1 = 0, ay0 = 0, ay1 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fma_c(ux, c_phi(q7x(q), q7y(q), i), u[i].x);
      fma_c(uy, c_phi(q7x(q), q7y(q), i), u[i].y);
      fma_c(ax0, c_dphi(q7x(q), q7y(q), i, 0), u[i].x);
      fma_c(ax1, c_dphi(q7x(q), q7y(q), i, 1), u[i].x);
      fma_c(ay0, c_dphi(q7x(q), q7y(q), i, 0), u[i].y);
      fma_c(ay1, c_dphi(q7x(q), q7y(q), i, 1), u[i].y);
    }
    const double uxx = g.j00 * ax0 + g.j10 * ax1, uxy = g.j01 * ax0 + g.j11 * ax1;
    const double uyx = g.j00 * ay0 + g.j10 * ay1, uyy = g.j01 * ay0 + g.j11 * ay1;
    const double pq = p[0] * c_lam(q7x(q), q7y(q), 0) + p[1] * c_lam(q7x(q), q7y(q), 1) + p[2] * c_lam(q7x(q), q7y(q), 2);
    const double w = q7w(q) * g.det;
    const double sx = w * (a * ux - rho * (ux * uxx + uy * uxy));
    const double sy = w * (a * uy - rho * (ux * uyx + uy * uyy));
    const double exy = 0.5 * (uxy + uyx);
    const double Fxx = w * (pq - mu * uxx), Fxy = -w * mu * exy, Fyy = w * (pq - mu * uyy);
    const double rx0 = g.j00 * Fxx + g.j01 * Fxy, rx1 = g.j10 * Fxx + g.j11 * Fxy;
    const double ry0 = g.j00 * Fxy + g.j01 * Fyy, ry1 = g.j10 * Fxy + g.j11 * Fyy;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fma_c(r[i].x, c_phi(q7x(q), q7y(q), i), sx);
      fma_c(r[i].x, c_dphi(q7x(q), q7y(q), i, 0), rx0);
      fma_c(r[i].x, c_dphi(q7x(q), q7y(q), i, 1), rx1);
      fma_c(r[i].y, c_phi(q7x(q), q7y(q), i), sy);
      fma_c(r[i].y, c_dphi(q7x(q), q7y(q), i, 0), ry0);
      fma_c(r[i].y, c_dphi(q7x(q), q7y(q), i, 1), ry1);
    }
  }
}

// step 2 (flow_solver.py:115-116): r_j = int grad p_n . grad psi_j - (1/dt) div(u*) psi_j
// div(u*) is linear on the cell: the edge-midpoint rule (degree 2) is exact.
__device__ __forceinline__ void elem_rhs2(const Geo& g, double idt, const double2 (&u)[6], const double (&p)[3],
                                          double (&r)[3]) {
  const double glx[3] = {-g.j00 - g.j10, g.j00, g.j10};
  const double gly[3] = {-g.j01 - g.j11, g.j01, g.j11};
  const double gpx = p[0] * glx[0] + p[1] * glx[1] + p[2] * glx[2];
  const double gpy = p[0] * gly[0] + p[1] * gly[1] + p[2] * gly[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) r[j] = 0.5 * g.det * (gpx * glx[j] + gpy * gly[j]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double ax0 = 0, ax1 = 0, ay0 = 0, ay1 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      fma_c(ax0, c_dphi(q3x(q), q3y(q), i, 0), u[i].x);
      fma_c(ax1, c_dphi(q3x(q), q3y(q), i, 1), u[i].x);
      fma_c(ay0, c_dphi(q3x(q), q3y(q), i, 0), u[i].y);
      fma_c(ay1, c_dphi(q3x(q), q3y(q), i, 1), u[i].y);
    }
    const double div = (g.j00 * ax0 + g.j10 * ax1) + (g.j01 * ay0 + g.j11 * ay1);
    const double w = g.det * (1.0 / 6.0) * idt * div;
#pragma unroll
    for (int j = 0; j < 3; ++j) fma_c(r[j], -c_lam(q3x(q), q3y(q), j), w);
  }
}

// step 3 (flow_solver.py:119-120): r^c_i = int (u*_c - dt d_c(p - p_n)) phi_i
//   = (M_e u*)_i^c - dt d_c(dp) int phi_i ;  int phi_i = 0 (vertices), det/6 (edges)
__device__ __forceinline__ void elem_rhs3(const Geo& g, double dt, const double2 (&u)[6], const double (&dp)[3],
                                          double2 (&r)[6]) {
  elem_mass(g, u, r);
  const double glx[3] = {-g.j00 - g.j10, g.j00, g.j10};
  const double gly[3] = {-g.j01 - g.j11, g.j01, g.j11};
  const double gx = dt * (dp[0] * glx[0] + dp[1] * glx[1] + dp[2] * glx[2]) * g.det * (1.0 / 6.0);
  const double gy = dt * (dp[0] * gly[0] + dp[1] * gly[1] + dp[2] * gly[2]) * g.det * (1.0 / 6.0);
#pragma unroll
  for (int i = 3; i < 6; ++i) {
    r[i].x -= gx;
    r[i].y -= gy;
  }
}

}  // namespace mdq
