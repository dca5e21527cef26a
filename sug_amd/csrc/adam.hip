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


// ---- several optimizers, one launch (sug_adam_chain_step) ------------------------------------------------------
// A SUG step runs three Adam optimizers back to back, and two of them own the encoder's parameters
// (train_dg_single_gpu.py:193-203: optimizer_dis and optimizer_g both hold model.g.parameters()): six launches of
// latency (a per-workgroup binary search, four dependent load -> store rounds) for ~100 MB of traffic.  Here every
// tensor carries the ordered list of its (moment, bucket) slots -- up to SUG_ADAM_SLOTS updates applied one after the
// other on the value held in registers, each exactly the arithmetic of adam_kernel -- so the parameter and its
// gradient are read once and written once, the workgroup finds its tensor in ONE load, and all loads of a workgroup
// are in flight before the first update.  Bit-identical to the optimizers stepped one after the other in that order.
#define SUG_ADAM_CHAIN_CHUNK 2048     // elements per workgroup
#define SUG_ADAM_SLOTS SUG_ADAM_CHAIN_SLOTS
#define SUG_ADAM_BUCKETS SUG_ADAM_CHAIN_BUCKETS

struct AdamConsts {
  float step_size, inv_bc2_sqrt, omb1, beta2, omb2, eps, wd, pad;
};
struct AdamChainConsts {
  AdamConsts c[SUG_ADAM_BUCKETS];
};
struct AdamPrepareArgs {
  double lr[SUG_ADAM_BUCKETS], beta1[SUG_ADAM_BUCKETS], beta2[SUG_ADAM_BUCKETS];
};

// table row (8 x int64): p, numel, m0, v0, m1, v1, nslots | bucket0 << 8 | bucket1 << 16, unused
__global__ __launch_bounds__(256) void adam_chain_kernel(const int64_t* __restrict__ table,
                                                         const int2* __restrict__ block_map,   // (tensor, chunk) per workgroup
                                                         int block0, int t0, GradPtrs gp, AdamChainConsts hc,
                                                         const float* __restrict__ dev_scalars) {
  const int2 bm = block_map[block0 + blockIdx.x];
  const float* __restrict__ g = gp.g[bm.x - t0];
  if (!g) return;
  const int64_t* row = table + 8 * (int64_t)bm.x;
  float* __restrict__ p = (float*)row[0];
  const int64_t n = row[1];
  const int code = (int)row[6];
  const int nslots = code & 255;
  float* mp[SUG_ADAM_SLOTS];
  float* vp[SUG_ADAM_SLOTS];
  AdamConsts hcs[SUG_ADAM_SLOTS];
#pragma unroll
  for (int s = 0; s < SUG_ADAM_SLOTS; ++s) {
    const int bk = (code >> (8 + 8 * s)) & 255;
    mp[s] = (float*)row[2 + 2 * s];
    vp[s] = (float*)row[3 + 2 * s];
    hcs[s] = hc.c[s < nslots ? bk : 0];
    if (dev_scalars && s < nslots) {              // capturable mode: written by adam_chain_prepare_kernel
      hcs[s].step_size = dev_scalars[2 * bk];
      hcs[s].inv_bc2_sqrt = dev_scalars[2 * bk + 1];
    }
  }
  auto upd = [](float& pw, float gw, float& mw, float& vw, const AdamConsts& h) {
    gw = gw + pw * h.wd;
    mw = mw + h.omb1 * (gw - mw);                 // lerp(exp_avg, grad, 1 - beta1)
    vw = h.beta2 * vw + h.omb2 * gw * gw;
    const float denom = sqrtf(vw) * h.inv_bc2_sqrt + h.eps;
    pw = pw - h.step_size * (mw / denom);
  };
  const int64_t base = (int64_t)bm.y * SUG_ADAM_CHAIN_CHUNK;
  uintptr_t al = (uintptr_t)p | (uintptr_t)g;
#pragma unroll
  for (int s = 0; s < SUG_ADAM_SLOTS; ++s)
    if (s < nslots) al |= (uintptr_t)mp[s] | (uintptr_t)vp[s];
  if ((al & 15) == 0 && base + SUG_ADAM_CHAIN_CHUNK <= n) {
    constexpr int IT = SUG_ADAM_CHAIN_CHUNK / 1024;
    float4 pw[IT], gw[IT], mw[SUG_ADAM_SLOTS][IT], vw[SUG_ADAM_SLOTS][IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t e = base + it * 1024 + threadIdx.x * 4;
      pw[it] = *(const float4*)(p + e);
      gw[it] = *(const float4*)(g + e);
#pragma unroll
      for (int s = 0; s < SUG_ADAM_SLOTS; ++s)
        if (s < nslots) {
          mw[s][it] = *(const float4*)(mp[s] + e);
          vw[s][it] = *(const float4*)(vp[s] + e);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t e = base + it * 1024 + threadIdx.x * 4;
#pragma unroll
      for (int s = 0; s < SUG_ADAM_SLOTS; ++s)
        if (s < nslots) {
          upd(pw[it].x, gw[it].x, mw[s][it].x, vw[s][it].x, hcs[s]);
          upd(pw[it].y, gw[it].y, mw[s][it].y, vw[s][it].y, hcs[s]);
          upd(pw[it].z, gw[it].z, mw[s][it].z, vw[s][it].z, hcs[s]);
          upd(pw[it].w, gw[it].w, mw[s][it].w, vw[s][it].w, hcs[s]);
          *(float4*)(mp[s] + e) = mw[s][it];
          *(float4*)(vp[s] + e) = vw[s][it];
        }
      *(float4*)(p + e) = pw[it];
    }
  } else {
    for (int64_t e = base + threadIdx.x; e < n && e < base + SUG_ADAM_CHAIN_CHUNK; e += 256) {
      float pw = p[e];
      const float gw = g[e];
#pragma unroll
      for (int s = 0; s < SUG_ADAM_SLOTS; ++s)
        if (s < nslots) {
          float mw = mp[s][e], vw = vp[s][e];
          upd(pw, gw, mw, vw, hcs[s]);
          mp[s][e] = mw;
          vp[s][e] = vw;
        }
      p[e] = pw;
    }
  }
}

// adam_prepare_kernel for every bucket of a chain: thread i advances step[i] and writes scalars[2i], scalars[2i+1]
__global__ void adam_chain_prepare_kernel(int32_t* __restrict__ step, int nb, AdamPrepareArgs a,
                                          const double* __restrict__ lr_dev, float* __restrict__ scalars) {
  const int i = threadIdx.x;
  if (i >= nb) return;
  const int t = step[i] + 1;
  step[i] = t;
  const double lr = lr_dev ? lr_dev[i] : a.lr[i];
  scalars[2 * i] = (float)(lr / (1.0 - pow(a.beta1[i], (double)t)));
  scalars[2 * i + 1] = (float)(1.0 / sqrt(1.0 - pow(a.beta2[i], (double)t)));
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

extern "C" int sug_adam_chain_chunk(void) { return SUG_ADAM_CHAIN_CHUNK; }

extern "C" int sug_adam_chain_step(const int64_t* table, const int32_t* block_map, const int32_t* block_first_host,
                                   int T, const void* const* grads_host, int nbuckets, const double* hyper_host,
                                   int32_t* steps_dev, float* scalars_dev, const double* lr_dev, void* stream) {
  SUG_REQUIRE(table && block_map && block_first_host && grads_host && hyper_host, "sug_adam_chain_step: null pointer");
  SUG_REQUIRE(T > 0 && nbuckets > 0 && nbuckets <= SUG_ADAM_BUCKETS, "sug_adam_chain_step: %d tensors, %d buckets (at most %d)",
              T, nbuckets, SUG_ADAM_BUCKETS);
  SUG_REQUIRE((steps_dev == nullptr) == (scalars_dev == nullptr), "sug_adam_chain_step: steps_dev and scalars_dev go together");
  hipStream_t st = (hipStream_t)stream;
  AdamChainConsts hc;
  AdamPrepareArgs pa;
  for (int i = 0; i < SUG_ADAM_BUCKETS; ++i) {
    const double* h = hyper_host + 6 * (i < nbuckets ? i : 0);      // lr, beta1, beta2, eps, weight decay, step count t
    const double lr = h[0], b1 = h[1], b2 = h[2], t = h[5];
    AdamConsts& c = hc.c[i];
    c.omb1 = (float)(1.0 - b1);
    c.beta2 = (float)b2;
    c.omb2 = (float)(1.0 - b2);
    c.eps = (float)h[3];
    c.wd = (float)h[4];
    c.pad = 0.f;
    c.step_size = c.inv_bc2_sqrt = 0.f;
    if (!steps_dev) {                                               // by value: the bias corrections of step t
      SUG_REQUIRE(t >= 1.0, "sug_adam_chain_step: bucket %d at step %g", i, t);
      c.step_size = (float)(lr / (1.0 - pow(b1, t)));
      c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(b2, t)));
    }
    pa.lr[i] = lr;
    pa.beta1[i] = b1;
    pa.beta2[i] = b2;
  }
  if (steps_dev) {
    hipLaunchKernelGGL(adam_chain_prepare_kernel, dim3(1), dim3(64), 0, st, steps_dev, nbuckets, pa, lr_dev, scalars_dev);
    SUG_LAUNCH_CHECK("sug_adam_chain_step(prepare)");
  }
  for (int t0 = 0; t0 < T; t0 += SUG_ADAM_ARGS) {
    const int tn = T - t0 < SUG_ADAM_ARGS ? T - t0 : SUG_ADAM_ARGS;
    GradPtrs gp;
    for (int i = 0; i < SUG_ADAM_ARGS; ++i) gp.g[i] = i < tn ? (const float*)grads_host[t0 + i] : nullptr;
    const int blocks = block_first_host[t0 + tn] - block_first_host[t0];
    if (blocks <= 0) continue;
    hipLaunchKernelGGL(adam_chain_kernel, dim3(blocks), dim3(256), 0, st, table, (const int2*)block_map,
                       block_first_host[t0], t0, gp, hc, (const float*)scalars_dev);
    SUG_LAUNCH_CHECK("sug_adam_chain_step");
  }
  return SUG_OK;
}
