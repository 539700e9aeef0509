// Table slabs of the large-mesh kernel instances (meshes beyond 1024 vertices keep their tables in global memory instead
// of LDS): ONE slab per (kernel family, device, stream), grown on demand and kept for the life of the process.  Keyed by
// the stream because two engines of a process may run their large-mesh kernels at the same time - the main stream's
// topology run and the flow stream's (VecEnv2DAirfoil with flow_overlap on a refined mesh), or two environment groups:
// launches on one stream are ordered and may share a slab, launches on different streams may not.  (Rounds 4's first
// version had one slab per family and process: correct only while a single stream used the large instances.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <map>
#include <mutex>
#include <utility>

namespace mdq_slab {

struct Pool {
  std::mutex m;
  std::map<std::pair<int, void*>, std::pair<unsigned char*, size_t>> slabs;   // (device, stream) -> (pointer, bytes)
  // the slab of this stream, at least `need` bytes; nullptr on failure
  unsigned char* get(void* stream, size_t need) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(m);
    auto& e = slabs[std::make_pair(dev, stream)];
    if (need > e.second) {
      if (e.first) {
        // only launches on THIS stream use the old slab: wait for them, then release it
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipFree(e.first) != hipSuccess) return nullptr;
        e.first = nullptr;
        e.second = 0;
      }
      if (hipMalloc(reinterpret_cast<void**>(&e.first), need) != hipSuccess) {
        e.first = nullptr;
        return nullptr;
      }
      e.second = need;
    }
    return e.first;
  }
};

}  // namespace mdq_slab
