// Single-wave instruction latency / issue microbenchmark (gfx950): cycles per instruction of dependent and independent
// chains of the instructions the smoothing pass is made of, and the LDS store -> load -> use round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__global__ void bench(double* out, long long* cyc, int n, unsigned mask) {
  __shared__ __attribute__((aligned(16))) double lds[1024];
  double a = out[threadIdx.x], b = out[64 + threadIdx.x], c = 1.0000001, d = a + 1;
  float f = (float)a, g = (float)b;
  int k = 0;
  unsigned long long t0, t1;
  // 0: dependent v_add_f64
  if (mask & (1u << 0)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_add_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(c));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  } else ++k;
  // 1: dependent v_fma_f64
  if (mask & (1u << 1)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_fma_f64 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(c));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  } else ++k;
  // 2: two independent v_add_f64 chains interleaved (128 instructions per rep)
  if (mask & (1u << 2)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2\n\t") : "+v"(a), "+v"(b) : "v"(c));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 2; ++k;
  } else ++k;
  // 3: four independent chains
  if (mask & (1u << 3)) {
  t0 = now();
  for (int i = 0; i < n; ++i)
    asm volatile(REP64("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t")
                 : "+v"(a), "+v"(b), "+v"(d), "+v"(c) : "v"(1.0));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 4; ++k;
  } else ++k;
  // 4: dependent v_add_f32
  if (mask & (1u << 4)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(f) : "v"(g));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  } else ++k;
  // 5: four independent v_add_f32
  if (mask & (1u << 5)) {
  {
    float f1 = f, f2 = g, f3 = f + 1, f4 = g + 1;
    t0 = now();
    for (int i = 0; i < n; ++i)
      asm volatile(REP64("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t")
                   : "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(1.0f));
    t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 4; ++k;
    f += f1 + f2 + f3 + f4;
  }
  } else ++k;
  // 6: dependent s_nop 1 + v_mov_b32_dpp + v_add_f32 (the 32-bit analogue of one exchange + add)
  if (mask & (1u << 6)) {
  {
    float t = g, x = f;
    t0 = now();
    for (int i = 0; i < n; ++i)
      asm volatile(REP64("s_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %0, %1\n\t")
                   : "+v"(x), "+v"(t));
    t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    f += x + t;
  }
  } else ++k;
  // 7: dependent v_cvt_f32_f64 + v_cvt_f64_f32 pair
  if (mask & (1u << 7)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %0, %1\n\t") : "+v"(a), "+v"(f));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 2; ++k;
  } else ++k;
  // 8: dependent v_rcp_f32
  if (mask & (1u << 8)) {
  t0 = now();
  for (int i = 0; i < n; ++i) asm volatile(REP64("v_rcp_f32 %0, %0\n\t") : "+v"(f));
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  } else ++k;
  // 9: dependent v_min_u32_dpp (with s_nop 1)
  if (mask & (1u << 9)) {
  {
    unsigned u = (unsigned)threadIdx.x * 77u;
    t0 = now();
    for (int i = 0; i < n; ++i) asm volatile(REP64("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(u));
    t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    f += (float)u;
  }
  } else ++k;
  // 10: LDS round trip: ds_write_b64 -> ds_read_b64 (same address) -> v_add_f64 on the loaded value -> write ...
  if (mask & (1u << 10)) {
  {
    unsigned addr = (unsigned)(threadIdx.x & 63) * 16;
    double v = b, w = a;
    t0 = now();
    for (int i = 0; i < n; ++i)
      asm volatile(REP16("ds_write_b64 %2, %1\n\tds_read_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 %1, %0, %1\n\t")
                   : "+v"(v), "+v"(w) : "v"(addr) : "memory");
    t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) * 4; ++k;   // per 64 round trips
    a += w + v;
  }
  } else ++k;
  // 11: ds_read_b128 alone, dependent address (pointer chase through LDS: value = own address)
  if (mask & (1u << 11)) {
  {
    unsigned addr = (unsigned)(threadIdx.x & 63) * 16;
    ((unsigned*)lds)[threadIdx.x * 4] = addr;
    __syncthreads();
    t0 = now();
    for (int i = 0; i < n; ++i) asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(addr)::"memory");
    t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) * 4; ++k;
    f += addr;
  }
  } else ++k;
  // 12: scalar dependent s_add_u32
  if (mask & (1u << 12)) {
  {
    unsigned s = n;
    t0 = now();
    for (int i = 0; i < n; ++i) asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s));
    t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    f += s;
  }
  } else ++k;
  // 13: v_cmp_lt_f32 -> s_and_b64 -> v_cndmask dependent triple
  if (mask & (1u << 13)) {
  {
    float x = f;
    t0 = now();
    for (int i = 0; i < n; ++i)
      asm volatile(REP64("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(x) : "v"(g) : "vcc");
    t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 3; ++k;
    f += x;
  }
  } else ++k;
  out[threadIdx.x] = a + b + c + d + f + lds[threadIdx.x];
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const unsigned mask = argc > 1 ? strtoul(argv[1], nullptr, 0) : 0xFFFFu;
  double* out; long long* cyc;
  hipMalloc(&out, 8 * 256); hipMemset(out, 0, 8 * 256);
  hipHostMalloc(&cyc, 8 * 32);
  const int n = 20;
  const char* names[] = {"dep v_add_f64", "dep v_fma_f64", "2 indep v_add_f64", "4 indep v_add_f64", "dep v_add_f32", "4 indep v_add_f32",
                         "dep s_nop1 + v_mov_b32_dpp + v_add_f32 (per triple)", "dep cvt f64<->f32 (per cvt)", "dep v_rcp_f32",
                         "dep s_nop1 + v_min_u32_dpp", "LDS write64 -> read128 -> add_f64 round trip", "LDS dependent ds_read_b32",
                         "dep s_add_u32", "v_cmp / s_and / v_cndmask (per instr)"};
  for (int blocks : {1, 2}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(bench, dim3(1), dim3(64 * blocks), 0, 0, out, cyc, n, mask); hipDeviceSynchronize(); }
    printf("waves on the CU: %d (on different SIMDs)\n", blocks);
    for (int k = 0; k < 14; ++k) printf("  %-60s %.2f cycles\n", names[k], (double)cyc[k] / (64.0 * n));
  }
  return 0;
}
