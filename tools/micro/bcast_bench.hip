// Microbenchmark for the block step of smooth_linear_kernel (VERDICT r4 #6b): how to hand the 32 right-hand sides g of a
// block to the lanes that multiply them - ONE wave, lane = row i | half h << 5, every lane needs the 16 values g[16 h ..
// 16 h + 15] of its half for two chains of 8 FMAs.  Variants (cycles per step by s_memtime, the steps form a dependent chain
// as in a sweep: the result of a step feeds the next one's g):
//   0  as the kernel does it: half 0 stores g to LDS, every lane reads its 16 values back as 8 ds_read_b128
//   1  v_readlane to scalar registers (2 per value: 64 readlanes), the half picked by v_cndmask, FMAs with register operands
//   2  ds_bpermute_b32 (2 per value: 32 per lane), no LDS memory
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/bcast_bench.hip -o tools/micro/bin/bcast_bench
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int REP = 2000;

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double halves_sum(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(const double* m, long long* cycles, double* sink) {
  __shared__ __attribute__((aligned(16))) double G[32];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  double M[16];
  for (int t = 0; t < 16; ++t) M[t] = m[lane * 16 + t];
  double g = 1.0 + 1e-3 * lane;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
    double acc0 = 0.0, acc1 = 0.0;
    if (MODE == 0) {
      if (h == 0) G[i] = g;
      asm volatile("" ::: "memory");
      const double2* gq = reinterpret_cast<const double2*>(G + 16 * h);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const double2 gg = gq[t];
        acc0 = __builtin_fma(M[2 * t], gg.x, acc0);
        acc1 = __builtin_fma(M[2 * t + 1], gg.y, acc1);
      }
      asm volatile("" ::: "memory");
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const double a0 = readlane_d(g, 2 * t), a1 = readlane_d(g, 2 * t + 1), b0 = readlane_d(g, 16 + 2 * t), b1 = readlane_d(g, 17 + 2 * t);
        acc0 = __builtin_fma(M[2 * t], h ? b0 : a0, acc0);
        acc1 = __builtin_fma(M[2 * t + 1], h ? b1 : a1, acc1);
      }
    } else {
      const int lo = __double2loint(g), hi = __double2hiint(g);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int s0 = (16 * h + 2 * t) * 4, s1 = s0 + 4;
        const double x0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(s0, hi), __builtin_amdgcn_ds_bpermute(s0, lo));
        const double x1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(s1, hi), __builtin_amdgcn_ds_bpermute(s1, lo));
        acc0 = __builtin_fma(M[2 * t], x0, acc0);
        acc1 = __builtin_fma(M[2 * t + 1], x1, acc1);
      }
    }
    g = halves_sum(acc0 + acc1) * 0.03 + 1.0;          // (the step's result feeds the next step: a dependent chain)
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[0] = t1 - t0;
  sink[lane] = g;
}

int main() {
  double *m, *sink;
  long long* cyc;
  hipMalloc(&m, 64 * 16 * 8); hipMalloc(&sink, 64 * 8); hipMalloc(&cyc, 8);
  double hm[64 * 16];
  for (int j = 0; j < 64 * 16; ++j) hm[j] = 0.01 + 1e-4 * j;
  hipMemcpy(m, hm, sizeof hm, hipMemcpyHostToDevice);
  const char* names[3] = {"LDS store + 8 ds_read_b128 (the kernel's form)", "64 v_readlane + v_cndmask", "32 ds_bpermute_b32"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, m, cyc, sink);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, m, cyc, sink);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, m, cyc, sink);
      hipDeviceSynchronize();
    }
    long long c;
    double s[64];
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(s, sink, sizeof s, hipMemcpyDeviceToHost);
    printf("variant %d  %-48s %7.1f s_memtime ticks per step (result %.6f)\n", mode, names[mode], (double)c / REP, s[0]);
  }
  return 0;
}
