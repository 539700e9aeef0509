// Is v_mfma_f32_{32x32x2,16x16x4}_f32 bitwise a sequential fmaf chain over k (k = 0, 1, ... within and across instructions)?
// hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_exact.hip -o tools/micro/bin/mfma_exact && tools/micro/bin/mfma_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int K = 256;
// A [32][K], B [K][32] -> D [32][32]
__global__ void k32(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  v16f acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    const float a = A[(l % 32) * K + k0 + l / 32];   // A operand: lane l holds A[m = l % 32][k = l / 32]
    const float b = B[(k0 + l / 32) * 32 + l % 32];  // B operand: lane l holds B[k = l / 32][n = l % 32]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // D: lane l, reg r: row = 8 * (r / 4) + 4 * (l / 32) ... standard layout: m = (r / 4) * 8 + (l / 32) * 4 + r % 4, n = l % 32
  for (int r = 0; r < 16; ++r) D[((r / 4) * 8 + (l / 32) * 4 + r % 4) * 32 + l % 32] = acc[r];
}
__global__ void k16(const float* A, const float* B, float* D) {   // A [16][K], B [K][16]
  const int l = threadIdx.x;
  v4f acc = {0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float a = A[(l % 16) * K + k0 + l / 16];
    const float b = B[(k0 + l / 16) * 16 + l % 16];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l / 16) * 4 + r) * 16 + l % 16] = acc[r];
}
int main() {
  const int M = 32;
  float *hA = (float*)malloc(M * K * 4), *hB = (float*)malloc(K * M * 4), *hD = (float*)malloc(M * M * 4);
  int bad32 = 0, bad16 = 0, bad32s = 0, trials = 200;
  float *A, *B, *D;
  hipMalloc(&A, M * K * 4); hipMalloc(&B, K * M * 4); hipMalloc(&D, M * M * 4);
  srand(1);
  for (int t = 0; t < trials; ++t) {
    const float sc = t % 3 == 0 ? 1e-3f : (t % 3 == 1 ? 1.f : 1e3f);
    for (int i = 0; i < M * K; ++i) { hA[i] = sc * ((float)rand() / RAND_MAX - 0.5f); hB[i] = ((float)rand() / RAND_MAX - 0.5f); }
    hipMemcpy(A, hA, M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, K * M * 4, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, M * M * 4, hipMemcpyDeviceToHost);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      float acc = 0.f, s = 0.f; for (int k = 0; k < K; ++k) { acc = fmaf(hA[m * K + k], hB[k * 32 + n], acc); s += hA[m * K + k] * hB[k * 32 + n]; }
      if (memcmp(&acc, &hD[m * 32 + n], 4)) ++bad32;
      if (memcmp(&s, &hD[m * 32 + n], 4)) ++bad32s;
    }
    // 16x16: reuse the first 16 rows of A (stride K) and a [K][16] B
    for (int i = 0; i < K * 16; ++i) hB[i] = ((float)rand() / RAND_MAX - 0.5f);
    hipMemcpy(B, hB, K * 16 * 4, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, 16 * 16 * 4, hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      float acc = 0.f; for (int k = 0; k < K; ++k) acc = fmaf(hA[m * K + k], hB[k * 16 + n], acc);
      if (memcmp(&acc, &hD[m * 16 + n], 4)) ++bad16;
    }
  }
  printf("32x32x2: %d of %d outputs differ from the fmaf chain (%d from mul+add chain); 16x16x4: %d of %d differ\n", bad32, trials * 1024, bad32s, bad16, trials * 256);
  return 0;
}
