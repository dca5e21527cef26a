// Multi-tensor Adam step (torch.optim.Adam semantics, train_dg_single_gpu.py:193-203: three Adam
// optimizers over ~150 small tensors).  One launch per optimizer instead of torch's ~10+10:
// parameter / moment pointers and sizes sit in a device table built once, the gradient pointers
// (fresh allocations every step) travel in the kernel arguments.  HBM-bound: 16 B read + 12 B
// written per element.
#include "common.h"

#define SUG_ADAM_CHUNK 4096           // elements per workgroup
#define SUG_ADAM_ARGS 384             // gradient pointers per launch (3 KB of kernarg)

namespace {

struct GradPtrs {
  const float* g[SUG_ADAM_ARGS];
};

__global__ __launch_bounds__(256) void adam_kernel(const int64_t* __restrict__ table,      // [T,4]: p, m, v, numel
                                                   const int32_t* __restrict__ block_first,  // [T+1]
                                                   int t0, int T, GradPtrs gp, float step_size, float omb1,
                                                   float beta2, float omb2, float eps, float wd,
                                                   float inv_bc2_sqrt, const float* __restrict__ dev_scalars) {
  if (dev_scalars) {                              // capturable mode: written by adam_prepare_kernel
    step_size = dev_scalars[0];
    inv_bc2_sqrt = dev_scalars[1];
  }
  const int b = blockIdx.x + block_first[t0];
  int lo = t0, hi = t0 + T;                       // block_first[lo] <= b < block_first[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (block_first[mid] <= b) lo = mid; else hi = mid;
  }
  const float* __restrict__ g = gp.g[lo - t0];
  if (!g) return;
  float* __restrict__ p = (float*)table[4 * lo + 0];
  float* __restrict__ m = (float*)table[4 * lo + 1];
  float* __restrict__ v = (float*)table[4 * lo + 2];
  const int64_t n = table[4 * lo + 3];
  const int64_t base = (int64_t)(b - block_first[lo]) * SUG_ADAM_CHUNK;
  auto upd = [&](float& pw, float gw, float& mw, float& vw) {
    gw = gw + pw * wd;
    mw = mw + omb1 * (gw - mw);                   // lerp(exp_avg, grad, 1 - beta1)
    vw = beta2 * vw + omb2 * gw * gw;
    const float denom = sqrtf(vw) * inv_bc2_sqrt + eps;
    pw = pw - step_size * (mw / denom);
  };
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec && base + SUG_ADAM_CHUNK <= n) {
#pragma unroll
    for (int it = 0; it < SUG_ADAM_CHUNK / 1024; ++it) {
      const int64_t e = base + it * 1024 + threadIdx.x * 4;
      float4 pw = *(const float4*)(p + e), mw = *(const float4*)(m + e), vw = *(const float4*)(v + e);
      const float4 gw = *(const float4*)(g + e);
      upd(pw.x, gw.x, mw.x, vw.x);
      upd(pw.y, gw.y, mw.y, vw.y);
      upd(pw.z, gw.z, mw.z, vw.z);
      upd(pw.w, gw.w, mw.w, vw.w);
      *(float4*)(p + e) = pw;
      *(float4*)(m + e) = mw;
      *(float4*)(v + e) = vw;
    }
  } else {
    for (int64_t e = base + threadIdx.x; e < n && e < base + SUG_ADAM_CHUNK; e += 256) {
      float pw = p[e], mw = m[e], vw = v[e];
      upd(pw, g[e], mw, vw);
      p[e] = pw;
      m[e] = mw;
      v[e] = vw;
    }
  }
}

// Capturable mode (hipGraph replay): the step count lives on the device; one thread advances it and
// derives lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) for the update kernel of the same replay.
// lr_dev (optional): the learning rate as a device-resident double, so that a schedule step changes a value in
// memory instead of a kernel argument baked into a captured graph.
__global__ void adam_prepare_kernel(int32_t* __restrict__ step, double lr, const double* __restrict__ lr_dev,
                                    double beta1, double beta2, float* __restrict__ scalars) {
  const int t = step[0] + 1;
  step[0] = t;
  if (lr_dev) lr = lr_dev[0];
  scalars[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
  scalars[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)t)));
}

}  // namespace

extern "C" int sug_adam_chunk(void) { return SUG_ADAM_CHUNK; }

extern "C" int sug_adam_step(const int64_t* table, const int32_t* block_first, const int32_t* block_first_host,
                             int T, const void* const* grads_host, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double bias_corr1, double bias_corr2, void* stream) {
  SUG_REQUIRE(table && block_first && block_first_host && grads_host, "sug_adam_step: null pointer");
  SUG_REQUIRE(T > 0 && bias_corr1 > 0 && bias_corr2 > 0, "sug_adam_step: bad arguments");
  const float step_size = (float)(lr / bias_corr1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bias_corr2));
  for (int t0 = 0; t0 < T; t0 += SUG_ADAM_ARGS) {
    const int tn = T - t0 < SUG_ADAM_ARGS ? T - t0 : SUG_ADAM_ARGS;
    GradPtrs gp;
    for (int i = 0; i < SUG_ADAM_ARGS; ++i) gp.g[i] = i < tn ? (const float*)grads_host[t0 + i] : nullptr;
    const int blocks = block_first_host[t0 + tn] - block_first_host[t0];
    if (blocks <= 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, block_first, t0, tn, gp,
                       step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay, inv_bc2_sqrt, (const float*)nullptr);
    SUG_LAUNCH_CHECK("sug_adam_step");
  }
  return SUG_OK;
}

extern "C" int sug_adam_step_capturable(const int64_t* table, const int32_t* block_first,
                                        const int32_t* block_first_host, int T, const void* const* grads_host,
                                        double lr, double beta1, double beta2, double eps, double weight_decay,
                                        int32_t* step_dev, float* scalars_dev, const double* lr_dev, void* stream) {
  SUG_REQUIRE(table && block_first && block_first_host && grads_host && step_dev && scalars_dev,
              "sug_adam_step_capturable: null pointer");
  SUG_REQUIRE(T > 0, "sug_adam_step_capturable: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, step_dev, lr, lr_dev, beta1, beta2, scalars_dev);
  SUG_LAUNCH_CHECK("sug_adam_step_capturable(prepare)");
  for (int t0 = 0; t0 < T; t0 += SUG_ADAM_ARGS) {
    const int tn = T - t0 < SUG_ADAM_ARGS ? T - t0 : SUG_ADAM_ARGS;
    GradPtrs gp;
    for (int i = 0; i < SUG_ADAM_ARGS; ++i) gp.g[i] = i < tn ? (const float*)grads_host[t0 + i] : nullptr;
    const int blocks = block_first_host[t0 + tn] - block_first_host[t0];
    if (blocks <= 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, table, block_first, t0, tn, gp, 0.f,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, 0.f,
                       (const float*)scalars_dev);
    SUG_LAUNCH_CHECK("sug_adam_step_capturable");
  }
  return SUG_OK;
}
