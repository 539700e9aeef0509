// Microbenchmark: LDS fp64 atomic-add and b128 gather throughput of one CU (768 threads, like at_velocity_kernel).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int TW = 768, N = 3322, REP = 64, PER = 12;

template <int MODE>
__global__ __launch_bounds__(TW) void k(const int* idx, long long* cycles, double* sink) {
  __shared__ double2 Y[N];
  __shared__ double2 X[N];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += TW) { Y[i] = make_double2(0, 0); X[i] = make_double2(i, 1); }
  int my[6];
  for (int j = 0; j < 6; ++j) my[j] = idx[(blockIdx.x * TW + tid) * 6 + j];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  double acc = 0.0;
  for (int r = 0; r < REP; ++r) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        unsafeAtomicAdd(reinterpret_cast<double*>(Y) + 2 * my[j], 1.0);
        unsafeAtomicAdd(reinterpret_cast<double*>(Y) + 2 * my[j] + 1, 2.0);
      }
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) { const double2 x = X[my[j]]; acc += x.x + x.y; }
    }
    if (MODE == 4) {   // 64-bit integer atomics
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        atomicAdd(reinterpret_cast<unsigned long long*>(Y) + 2 * my[j], 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(Y) + 2 * my[j] + 1, 2ull);
      }
    }
    if (MODE == 5) {   // 32-bit float atomics
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        unsafeAtomicAdd(reinterpret_cast<float*>(Y) + 4 * my[j], 1.0f);
        unsafeAtomicAdd(reinterpret_cast<float*>(Y) + 4 * my[j] + 2, 2.0f);
      }
    }
    if (MODE == 6) {   // 32-bit integer atomics
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        atomicAdd(reinterpret_cast<unsigned*>(Y) + 4 * my[j], 1u);
        atomicAdd(reinterpret_cast<unsigned*>(Y) + 4 * my[j] + 2, 2u);
      }
    }
    if (MODE == 3) {   // plain (non-atomic) b128 read-modify-write, for comparison
#pragma unroll
      for (int j = 0; j < 6; ++j) { double2 y = Y[my[j]]; y.x += 1.0; y.y += 2.0; Y[my[j]] = y; }
    }
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) sink[0] = acc + Y[tid].x;
}

int main() {
  const int B = 128;
  std::vector<int> h(B * TW * 6);
  for (int pat = 0; pat < 3; ++pat) {
    srand(1);
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < TW; ++t)
        for (int j = 0; j < 6; ++j) {
          int v;
          if (pat == 0) v = (t * 6 + j) % N;                      // dense, conflict-free banks within a wave
          else if (pat == 1) v = rand() % N;                      // random
          else v = ((t * 4 + j * 131 + (rand() % 16)) % N);       // mesh-like: near-diagonal with jitter
          h[(b * TW + t) * 6 + j] = v;
        }
    int* d; long long* c; double* s;
    hipMalloc(&d, h.size() * 4); hipMalloc(&c, B * 8); hipMalloc(&s, 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[7] = {"atomics(12 f64/thread)", "gathers(6 b128/thread)", "both", "plain b128 rmw", "12 u64 atomics", "12 f32 atomics", "12 u32 atomics"};
    for (int mode = 0; mode < 7; ++mode) {
      for (int w = 0; w < 2; ++w) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(B), dim3(TW), 0, 0, d, c, s);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(B), dim3(TW), 0, 0, d, c, s);
      }
      hipDeviceSynchronize();
      long long hc[B];
      hipMemcpy(hc, c, B * 8, hipMemcpyDeviceToHost);
      double m = 0; for (int b = 0; b < B; ++b) m += hc[b]; m /= B;
      printf("pattern %d %-24s %8.0f cycles per round of %d threads (%.1f cycles per wave-instruction group)\n", pat, names[mode],
             m / REP, TW, m / REP / (TW / 64));
    }
    hipFree(d); hipFree(c); hipFree(s);
  }
  return 0;
}
