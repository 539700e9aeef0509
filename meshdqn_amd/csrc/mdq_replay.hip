// Device-resident replay memory and optimiser update of the DQN learning loop: with these three kernels a training
// step needs no host round trip and no per-transition Python object -
//   replay_step_kernel   : the B transitions of one batched env step into the record ring
//                          (ReplayMemory.push, airfoil_dqn.py:56-61, for B environments at once)
//   replay_sample_kernel : a minibatch of records -> the arrays of mdq_gcn_forward / mdq_gcn_train_step
//                          (ReplayMemory.sample + the Batch.from_data_list of DataWorker._get_data, airfoil_dqn.py:63-64,240-262)
//   adam_kernel          : torch.optim.Adam.step (weight decay as L2 term, bias corrections) on every trained
//                          parameter in one launch (ParameterServer.apply_gradients, airfoil_dqn.py:184-200)
// Record layout (float32, the layout of trainer.pack_transitions: what the ranks all-gather when the replay is shared):
//   [ x(s) N*F | x(s') N*F | src(s) EM | dst(s) EM | src(s') EM | dst(s') EM | edges(s) | edges(s') | action | reward | done ]
// (a terminal transition has zeros for s').
#include <hip/hip_runtime.h>

#include "../../include/meshdqn_hip.h"
#include "mdq_internal.h"

namespace mdq_replay {

__global__ __launch_bounds__(256) void replay_step_kernel(float* R, int rec_len, int capacity, int nf, int EM, const float* x,
                                                           const int32_t* es, const int32_t* ed, const int32_t* nedges,
                                                           int base_cur, int base_prev, const int32_t* act,
                                                           const double* rew, const uint8_t* done) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * nf;
  const int32_t* sb = es + (size_t)b * EM;
  const int32_t* db = ed + (size_t)b * EM;
  const int cnt = nedges[b];
  if (base_cur >= 0) {   // the state the next transition of environment b starts from
    float* r = R + (size_t)((base_cur + b) % capacity) * rec_len;
    for (int i = tid; i < nf; i += 256) r[i] = xb[i];
    for (int e = tid; e < EM; e += 256) {
      r[2 * nf + e] = e < cnt ? (float)sb[e] : 0.f;
      r[2 * nf + EM + e] = e < cnt ? (float)db[e] : 0.f;
    }
    if (tid == 0) r[2 * nf + 4 * EM] = (float)cnt;
  }
  if (base_prev >= 0) {  // ... and the state the previous one ended in (nothing if that one was terminal: the
    float* r = R + (size_t)((base_prev + b) % capacity) * rec_len;   // environment has been reset in place since)
    const bool dn = done[b] != 0;
    for (int i = tid; i < nf; i += 256) r[nf + i] = dn ? 0.f : xb[i];
    for (int e = tid; e < EM; e += 256) {
      const bool live = !dn && e < cnt;
      r[2 * nf + 2 * EM + e] = live ? (float)sb[e] : 0.f;
      r[2 * nf + 3 * EM + e] = live ? (float)db[e] : 0.f;
    }
    if (tid == 0) {
      float* t = r + 2 * nf + 4 * EM;
      t[1] = dn ? 0.f : (float)cnt;
      t[2] = (float)act[b];
      t[3] = (float)rew[b];
      t[4] = dn ? 1.f : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void replay_sample_kernel(mdq_replay_sample_desc D) {
  const int i = blockIdx.x, tid = threadIdx.x, nf = D.nf, EM = D.EM, tail = 2 * nf + 4 * EM;
  __shared__ int red[2][256];
  // packed edge offsets of this transition's two graphs: counts of the transitions before it
  int c0 = 0, c1 = 0;
  for (int j = tid; j < i; j += 256) {
    const float* t = D.R + (size_t)D.idx[j] * D.rec_len + tail;
    c0 += (int)t[0];
    c1 += t[4] > 0.5f ? (int)t[0] : (int)t[1];     // terminal: its own state as a masked placeholder
  }
  red[0][tid] = c0;
  red[1][tid] = c1;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      red[0][tid] += red[0][tid + off];
      red[1][tid] += red[1][tid + off];
    }
    __syncthreads();
  }
  const int o0 = red[0][0], o1 = red[1][0];
  const float* r = D.R + (size_t)D.idx[i] * D.rec_len;
  const float* t = r + tail;
  const bool dn = t[4] > 0.5f;
  const int n0 = (int)t[0], n1 = dn ? n0 : (int)t[1];
  const float* xn = dn ? r : r + nf;
  const float* sn = dn ? r + 2 * nf : r + 2 * nf + 2 * EM;
  for (int k = tid; k < nf; k += 256) {
    D.x_s[(size_t)i * nf + k] = r[k];
    D.x_n[(size_t)i * nf + k] = xn[k];
  }
  for (int e = tid; e < n0; e += 256) {
    D.esrc_s[o0 + e] = (int32_t)r[2 * nf + e];
    D.edst_s[o0 + e] = (int32_t)r[2 * nf + EM + e];
  }
  for (int e = tid; e < n1; e += 256) {
    D.esrc_n[o1 + e] = (int32_t)sn[e];
    D.edst_n[o1 + e] = (int32_t)sn[EM + e];
  }
  if (tid == 0) {
    D.edge_ptr_s[i] = o0;
    D.edge_ptr_n[i] = o1;
    if (i == (int)gridDim.x - 1) {
      D.edge_ptr_s[i + 1] = o0 + n0;
      D.edge_ptr_n[i + 1] = o1 + n1;
    }
    D.action[i] = (int64_t)t[2];
    D.reward[i] = t[3];
    D.nonfinal[i] = dn ? 0.f : 1.f;
  }
}

// torch.optim.Adam (amsgrad False, maximize False), one thread per parameter element of every segment
__global__ __launch_bounds__(256) void adam_kernel(mdq_adam_desc D) {
  const int s = blockIdx.y;
  if (s >= D.n) return;
  float* p = D.param[s];
  const int len = D.len[s], off = D.offset[s];
  const float lr_c = (float)(D.lr / D.bias_correction1), bc2s = (float)sqrt(D.bias_correction2);
  const float b1 = (float)D.beta1, b2 = (float)D.beta2, eps = (float)D.eps, wd = (float)D.weight_decay;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) {
    float g = D.grad[off + i];
    const float w = p[i];
    if (wd != 0.f) g = fmaf(wd, w, g);
    const float m = D.exp_avg[off + i] + (1.f - b1) * (g - D.exp_avg[off + i]);      // lerp, as torch does
    const float v = b2 * D.exp_avg_sq[off + i] + (1.f - b2) * (g * g);
    D.exp_avg[off + i] = m;
    D.exp_avg_sq[off + i] = v;
    p[i] = w - lr_c * (m / (sqrtf(v) / bc2s + eps));
  }
}

}  // namespace mdq_replay

extern "C" int mdq_replay_step(float* ring, int32_t rec_len, int32_t capacity, int32_t B, int32_t nf, int32_t EM,
                               const float* x, const int32_t* edge_src, const int32_t* edge_dst, const int32_t* nedges,
                               int32_t base_cur, int32_t base_prev, const int32_t* action, const double* reward,
                               const uint8_t* done, void* stream) {
  if (!ring || B <= 0 || !x || !edge_src || !edge_dst || !nedges || rec_len != 2 * nf + 4 * EM + 5 || capacity <= 0 ||
      (base_prev >= 0 && (!action || !reward || !done)))
    return mdq_set_error("mdq_replay_step: bad arguments");
  hipLaunchKernelGGL(mdq_replay::replay_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ring, rec_len, capacity, nf,
                     EM, x, edge_src, edge_dst, nedges, base_cur, base_prev, action, reward, done);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("replay_step_kernel launch failed");
  return 0;
}

extern "C" int mdq_replay_sample(const mdq_replay_sample_desc* d, void* stream) {
  if (!d || d->n <= 0 || !d->R || !d->idx || d->rec_len != 2 * d->nf + 4 * d->EM + 5 || !d->x_s || !d->x_n || !d->esrc_s ||
      !d->edst_s || !d->esrc_n || !d->edst_n || !d->edge_ptr_s || !d->edge_ptr_n || !d->action || !d->reward || !d->nonfinal)
    return mdq_set_error("mdq_replay_sample: bad arguments");
  hipLaunchKernelGGL(mdq_replay::replay_sample_kernel, dim3(d->n), dim3(256), 0, (hipStream_t)stream, *d);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("replay_sample_kernel launch failed");
  return 0;
}

extern "C" int mdq_adam_step(const mdq_adam_desc* d, void* stream) {
  if (!d || d->n <= 0 || d->n > MDQ_GCN_PACK_MAX || !d->grad || !d->exp_avg || !d->exp_avg_sq)
    return mdq_set_error("mdq_adam_step: bad arguments");
  hipLaunchKernelGGL(mdq_replay::adam_kernel, dim3(32, d->n), dim3(256), 0, (hipStream_t)stream, *d);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("adam_kernel launch failed");
  return 0;
}

// ---------------------------------------------------------------- stream-concurrency probe
// A kernel that keeps the workgroup dispatcher of its hardware queue busy: `wgs` workgroups that each hold `lds` bytes of
// LDS (one per CU at 150 KB) and spin for `ticks` of the 100 MHz wall clock.  meshdqn_amd/streams.py launches it with
// more workgroups than CUs on one stream and a tiny kernel on another: the tiny kernel finishes at once only if the
// two streams sit on hardware queues that dispatch independently.
namespace mdq_replay {
__global__ __launch_bounds__(64) void spin_kernel(long long ticks, int* sink) {
  extern __shared__ int spin_lds[];
  const long long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < ticks) acc += 1;
  spin_lds[threadIdx.x] = acc;
  if (sink && acc < 0) sink[0] = spin_lds[0];
}
}  // namespace mdq_replay

extern "C" int mdq_spin(int32_t wgs, int32_t lds_bytes, int64_t ticks_100mhz, void* stream) {
  if (wgs <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024) return mdq_set_error("mdq_spin: bad arguments");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mdq_replay::spin_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  hipLaunchKernelGGL(mdq_replay::spin_kernel, dim3(wgs), dim3(64), lds_bytes, (hipStream_t)stream, (long long)ticks_100mhz,
                     (int*)nullptr);
  if (hipGetLastError() != hipSuccess) return mdq_set_error("spin_kernel launch failed");
  return 0;
}

// A stream whose kernels may only be placed on the compute units of `mask` (bit i of word i / 32 = compute unit i in the
// driver's numbering: on a multi-XCD part consecutive bits go round-robin over the XCDs, then over the shader engines of
// an XCD, so "the first half of the bits" is half of every shader engine of every XCD).  meshdqn_amd/streams.py gives the
// main chain and the flow leg of the env step disjoint halves of the chip: both are chains of one-workgroup-per-
// environment kernels (128 workgroups that each want a compute unit of their own), and left to the dispatcher the
// workgroups of the two chains are packed onto shared compute units whenever their LDS fits.
extern "C" int mdq_stream_create_cu_mask(const uint32_t* mask, int32_t nwords, void** stream) {
  if (!mask || nwords <= 0 || !stream) return mdq_set_error("mdq_stream_create_cu_mask: bad arguments");
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  *stream = s;
  return 0;
}

extern "C" int mdq_stream_destroy(void* stream) {
  if (!stream) return 0;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return mdq_set_error(hipGetErrorString(e));
  return 0;
}
