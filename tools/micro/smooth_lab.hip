// Stand-alone lab for the smoothing kernel (no torch): times mdq_smooth on B copies of a mesh and checks it against a
// plain sequential host loop.  Build: hipcc -O3 --offload-arch=gfx950 [-DMDQ_SMOOTH_...] -DSMOOTH_SRC='"<path>"'
//   tools/micro/smooth_lab.hip -o smooth_lab ; run: ./smooth_lab mesh.bin [B] [reps]
// mesh.bin (tools/micro/make_mesh_bins.py): int32 nv, nt; double coords[nv][2]; int32 cells[nt][3] (ascending ids per cell)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/meshdqn_hip.h"
static int mdq_set_error(const char* m) {
  fprintf(stderr, "error: %s\n", m);
  return -1;
}
#ifndef SMOOTH_SRC
#define SMOOTH_SRC "../../meshdqn_amd/csrc/mdq_smooth.hip"
#endif
#include SMOOTH_SRC
#ifdef SMOOTH_FAST
#include "../../meshdqn_amd/csrc/mdq_smooth_linear.hip"
#endif

static void host_smooth(std::vector<double>& x, const std::vector<int>& tri, int nv, int nt, int iters) {
  std::vector<std::vector<int>> vc(nv), nb(nv);
  std::vector<int> seen(nv, 0);
  for (int t = 0; t < nt; ++t)
    for (int k = 0; k < 3; ++k) vc[tri[3 * t + k]].push_back(t);
  std::vector<char> onb(nv, 0);
  for (int v = 0; v < nv; ++v) {
    std::vector<int> cntn;
    for (int t : vc[v])
      for (int k = 0; k < 3; ++k) {
        const int w = tri[3 * t + k];
        if (w != v) cntn.push_back(w);
      }
    std::sort(cntn.begin(), cntn.end());
    for (size_t i = 0; i < cntn.size();) {
      size_t j = i;
      while (j < cntn.size() && cntn[j] == cntn[i]) ++j;
      if (j - i != 2) onb[v] = 1;
      nb[v].push_back(cntn[i]);
      i = j;
    }
    if (vc[v].empty()) onb[v] = 1;
  }
  for (int it = 0; it < iters; ++it)
    for (int v = 0; v < nv; ++v) {
      if (onb[v]) continue;
      const double px = x[2 * v], py = x[2 * v + 1];
      double sx = 0, sy = 0;
      for (int w : nb[v]) {
        sx += x[2 * w];
        sy += x[2 * w + 1];
      }
      sx /= nb[v].size();
      sy /= nb[v].size();
      double rmin = 0;
      for (int t : vc[v]) {
        int o[2], n = 0;
        for (int k = 0; k < 3; ++k)
          if (tri[3 * t + k] != v) o[n++] = tri[3 * t + k];
        const double ax = x[2 * o[0]], ay = x[2 * o[0] + 1], bx = x[2 * o[1]], by = x[2 * o[1] + 1];
        const double tx = bx - ax, ty = by - ay, len = std::sqrt(tx * tx + ty * ty);
        const double r = std::fabs((ty / len) * (px - ax) + (-tx / len) * (py - ay));
        rmin = rmin == 0 ? r : std::min(rmin, r);
      }
      const double dx = sx - px, dy = sy - py, r = std::sqrt(dx * dx + dy * dy);
      if (r < 3.0e-16) continue;
      const double step = std::min(0.5 * rmin, r);
      x[2 * v] = px + step * dx / r;
      x[2 * v + 1] = py + step * dy / r;
    }
}

#define CK(e)                                                                  \
  do {                                                                         \
    hipError_t _r = (e);                                                       \
    if (_r != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_r)); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  const int B = argc > 2 ? atoi(argv[2]) : 128, reps = argc > 3 ? atoi(argv[3]) : 10, iters = argc > 4 ? atoi(argv[4]) : 50;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int nv, nt;
  if (fread(&nv, 4, 1, f) != 1 || fread(&nt, 4, 1, f) != 1) return 2;
  std::vector<double> x0(2 * nv);
  std::vector<int> tri(3 * nt);
  if (fread(x0.data(), 8, 2 * nv, f) != (size_t)(2 * nv) || fread(tri.data(), 4, 3 * nt, f) != (size_t)(3 * nt)) return 2;
  fclose(f);
  std::vector<double> ref = x0;
  host_smooth(ref, tri, nv, nt, iters);
  const int NV = nv, NT = nt;
  double *dx, *dx0;
  int *dt, *dnv, *dnt, *dit;
  CK(hipMalloc(&dx, sizeof(double) * 2 * NV * B));
  CK(hipMalloc(&dx0, sizeof(double) * 2 * NV * B));
  CK(hipMalloc(&dt, sizeof(int) * 3 * NT * B));
  CK(hipMalloc(&dnv, 4 * B));
  CK(hipMalloc(&dnt, 4 * B));
  CK(hipMalloc(&dit, 4 * B));
  std::vector<int> hnv(B, nv), hnt(B, nt), hit(B, iters);
  for (int b = 0; b < B; ++b) {
    CK(hipMemcpy(dx0 + (size_t)b * 2 * NV, x0.data(), sizeof(double) * 2 * nv, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt + (size_t)b * 3 * NT, tri.data(), sizeof(int) * 3 * nt, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(dnv, hnv.data(), 4 * B, hipMemcpyHostToDevice));
  CK(hipMemcpy(dnt, hnt.data(), 4 * B, hipMemcpyHostToDevice));
  CK(hipMemcpy(dit, hit.data(), 4 * B, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<double> out((size_t)2 * NV * B), first;
  double worst = 0, tmin = 1e9, tsum = 0;
  bool same = true;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemcpy(dx, dx0, sizeof(double) * 2 * NV * B, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
#ifdef SMOOTH_FAST
    static void* ws = nullptr;
    const int64_t wsb = mdq_smooth_fast_workspace_bytes(B, NV);
    if (!ws) CK(hipMalloc(&ws, wsb));
    if (getenv("SMOOTH_PARTS")) {   // the three launches of mdq_smooth_fast one by one, timed
      const int64_t blocks = (NV + mdq_smooth_lin::BS - 1) / mdq_smooth_lin::BS + 2, mstride = blocks * mdq_smooth_lin::MBLK;
      double* mws = reinterpret_cast<double*>(ws);
      int32_t* redo = reinterpret_cast<int32_t*>(mws + (int64_t)B * mstride);
      hipEvent_t ev[3];
      for (auto& e : ev) CK(hipEventCreate(&e));
      CK(hipEventRecord(ev[0], 0));
      hipLaunchKernelGGL(mdq_smooth_lin::smooth_linear_kernel, dim3(B), dim3(mdq_smooth_lin::LWG), 0, 0, NV, NT, dx, dt, dnv, dnt, dit, nullptr, nullptr, 0, mws, mstride, redo, redo + B);
      CK(hipEventRecord(ev[1], 0));
      hipLaunchKernelGGL(mdq_smoothing::smooth_kernel, dim3(1), dim3(mdq_smoothing::SWG), 0, 0, B, NV, NT, dx, dt, dnv, dnt, redo, 0, nullptr);
      CK(hipEventRecord(ev[2], 0));
      CK(hipDeviceSynchronize());
      float b_, c;
      CK(hipEventElapsedTime(&b_, ev[0], ev[1]));
      CK(hipEventElapsedTime(&c, ev[1], ev[2]));
      std::vector<int> hr(B);
      CK(hipMemcpy(hr.data(), redo, 4 * B, hipMemcpyDeviceToHost));
      int nr = 0;
      for (int v : hr) nr += v > 0;
      if (rep == reps - 1) printf("  parts: linear %.3f ms, redo %.3f ms (%d envs handed back)\n", b_, c, nr);
    } else if (mdq_smooth_fast(B, NV, NT, dx, dt, dnv, dnt, dit, ws, wsb, nullptr)) return 3;
#else
    if (mdq_smooth(B, NV, NT, dx, dt, dnv, dnt, dit, nullptr)) return 3;
#endif
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) {
      tmin = std::min(tmin, (double)ms);
      tsum += ms;
    }
    CK(hipMemcpy(out.data(), dx, sizeof(double) * 2 * NV * B, hipMemcpyDeviceToHost));
    if (rep == 0) first.assign(out.begin(), out.begin() + 2 * nv);
    for (int b = 0; b < B; ++b)
      if (memcmp(out.data() + (size_t)b * 2 * NV, first.data(), sizeof(double) * 2 * nv)) same = false;
    for (int i = 0; i < 2 * nv; ++i) worst = std::max(worst, std::fabs(out[i] - ref[i]));
  }
#ifdef MDQ_SMOOTH_TRACE
  {  // timestamps (ready, done) of every update of environment 0 of the last launch: [64][1024][2] int64
    const char* tp = getenv("SMOOTH_TRACE_OUT");
    FILE* tf = fopen(tp ? tp : "smooth_trace.bin", "wb");
    fwrite(mdq_smooth_trace_host(), sizeof(long long), 2 * 64 * 1024, tf);
    fclose(tf);
    const long long* ph = mdq_smooth_trace_host() + 2 * 63 * 1024;
    printf("setup phases (cycles):");
    for (int i = 1; i < 10 && ph[2 * i]; ++i) printf(" %lld", ph[2 * i] - ph[2 * (i - 1)]);
    printf("\n");
  }
#endif
  printf("%s B=%d iters=%d: min %.3f ms mean %.3f ms | max|gpu-host| %.3e | bitwise identical over envs and launches: %s\n",
         argv[1], B, iters, tmin, tsum / std::max(1, reps - 1), worst, same ? "yes" : "NO");
  return worst < 1e-12 && same ? 0 : 4;
}
