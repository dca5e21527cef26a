// The scalar tail of a SUG step (train_dg_single_gpu.py:269-292, :314-324) as two launches each way instead of ~25:
//   * cross entropy of BOTH classifier heads on the source rows of the paired logits:
//       loss_cls = w * (CE(pred_s1, label) + CE(pred_s2, label)),  CE = nn.CrossEntropyLoss() (mean over the batch),
//     w = 0.5 * SRC_LOSS_WEIGHT * CLS_WEIGHT folded on the host; forward: log-softmax + NLL + mean + weight in one
//     single-workgroup kernel (torch: 2 x log_softmax, 2 x nll_loss, add, mul); backward: w * g * (softmax - onehot) / M
//     for the source rows and ZEROS for the target rows of the paired [2B, C] logits, so that no torch.stack / zero fill
//     rebuilds the pair's gradient;
//   * total = loss_cls + wg * v_geo + ws * (v_sem1 + v_sem2) with the reported parts wg*v_geo and ws*(v_sem1+v_sem2)
//     (the five scalar launches of the loss sum and their backward).
// fp32, fixed summation order (one workgroup; rows summed by thread 0 in order).
#include "common.h"

namespace {
constexpr int CE_MAXC = 32;       // classes
constexpr int CE_MAXROWS = 512;   // source rows x 2 heads

__global__ __launch_bounds__(512) void ce_pair_fwd_kernel(const float* __restrict__ l1, const float* __restrict__ l2, int64_t ld,
                                                          const int64_t* __restrict__ label, int M, int C, float w,
                                                          int64_t ignore_index, float* __restrict__ loss,
                                                          float* __restrict__ lse) {
  __shared__ float s_nll[CE_MAXROWS];
  const int t = threadIdx.x;
  if (t < 2 * M) {
    const int hd = t / M, i = t % M;
    const float* row = (hd ? l2 : l1) + (int64_t)i * ld;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - mx);
    const float l = mx + logf(s);                    // logsumexp of the row
    lse[t] = l;
    // nn.CrossEntropyLoss semantics: a row whose label is `ignore_index` contributes nothing and does not count in the
    // mean; any other label outside [0, C) is an error -- torch raises (a device-side assert); a kernel cannot, so the row
    // poisons the loss with NaN instead of being scored as some class (round 4 clamped such labels silently)
    const int64_t y = label[i];
    s_nll[t] = (y == ignore_index) ? 0.f : ((y < 0 || y >= C) ? NAN : l - row[y]);       // -log_softmax(row)[y]
  }
  __syncthreads();
  if (t == 0) {
    float a = 0.f, b = 0.f;
    int cnt = 0;
    for (int i = 0; i < M; ++i) cnt += label[i] != ignore_index;
    for (int i = 0; i < M; ++i) a += s_nll[i];
    for (int i = 0; i < M; ++i) b += s_nll[M + i];
    lse[2 * M] = (float)cnt;                         // rows that count: the backward divides by it (0 rows -> NaN, as torch)
    loss[0] = w * (a / (float)cnt + b / (float)cnt);
  }
}

__global__ __launch_bounds__(256) void ce_pair_bwd_kernel(const float* __restrict__ l1, const float* __restrict__ l2, int64_t ld,
                                                          const int64_t* __restrict__ label, int M, int Mtot, int C, float w,
                                                          int64_t ignore_index, const float* __restrict__ g,
                                                          const float* __restrict__ lse, float* __restrict__ d1,
                                                          float* __restrict__ d2) {
  const int64_t n = (int64_t)2 * Mtot * C;
  const float f = w * g[0] / lse[2 * M];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int hd = (int)(e / ((int64_t)Mtot * C));
    const int r = (int)((e / C) % Mtot), c = (int)(e % C);
    float v = 0.f;
    if (r < M) {
      const float x = (hd ? l2 : l1)[(int64_t)r * ld + c];
      const int64_t y = label[r];
      v = (y == ignore_index) ? 0.f : ((y < 0 || y >= C) ? NAN : f * (expf(x - lse[hd * M + r]) - (c == y ? 1.f : 0.f)));
    }
    (hd ? d2 : d1)[(int64_t)r * C + c] = v;
  }
}

__global__ void loss_combine_fwd_kernel(const float* __restrict__ lcls, const float* __restrict__ v0, const float* __restrict__ v1,
                                        const float* __restrict__ v2, float wg, float ws, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    const float geo = v0 ? wg * v0[0] : 0.f;
    const float sem = v1 ? ws * (v1[0] + (v2 ? v2[0] : 0.f)) : 0.f;
    float tot = lcls[0];
    if (v0) tot += geo;                              // (loss_cls + loss_geo) + loss_sem, the step's own order
    if (v1) tot += sem;
    out[0] = tot; out[1] = geo; out[2] = sem;
  }
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, float wg, float ws, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    const float gv = g[0];
    out[0] = gv; out[1] = wg * gv; out[2] = ws * gv; out[3] = ws * gv;
  }
}
}  // namespace

extern "C" int sug_ce_pair_fwd(const float* logits1, const float* logits2, int64_t ld, const int64_t* label, int M, int C,
                               float w, int64_t ignore_index, float* loss, float* lse, void* stream) {
  SUG_REQUIRE(logits1 && logits2 && label && loss && lse, "sug_ce_pair_fwd: null pointer");
  SUG_REQUIRE(M > 0 && 2 * M <= CE_MAXROWS && C > 0 && C <= CE_MAXC && ld >= C, "sug_ce_pair_fwd: M=%d rows, C=%d classes (2M <= %d, C <= %d)",
              M, C, CE_MAXROWS, CE_MAXC);
  hipLaunchKernelGGL(ce_pair_fwd_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, logits1, logits2, ld, label, M, C, w, ignore_index, loss, lse);
  SUG_LAUNCH_CHECK("sug_ce_pair_fwd");
  return SUG_OK;
}

extern "C" int sug_ce_pair_bwd(const float* logits1, const float* logits2, int64_t ld, const int64_t* label, int M, int Mtot,
                               int C, float w, int64_t ignore_index, const float* g, const float* lse, float* d1, float* d2,
                               void* stream) {
  SUG_REQUIRE(logits1 && logits2 && label && g && lse && d1 && d2, "sug_ce_pair_bwd: null pointer");
  SUG_REQUIRE(M > 0 && Mtot >= M && 2 * M <= CE_MAXROWS && C > 0 && C <= CE_MAXC && ld >= C, "sug_ce_pair_bwd: bad shape");
  const int64_t n = (int64_t)2 * Mtot * C;
  hipLaunchKernelGGL(ce_pair_bwd_kernel, dim3((unsigned)sug_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, logits1, logits2, ld,
                     label, M, Mtot, C, w, ignore_index, g, lse, d1, d2);
  SUG_LAUNCH_CHECK("sug_ce_pair_bwd");
  return SUG_OK;
}

extern "C" int sug_loss_combine_fwd(const float* loss_cls, const float* v_geo, const float* v_sem1, const float* v_sem2, float wg,
                                    float ws, float* out3, void* stream) {
  SUG_REQUIRE(loss_cls && out3, "sug_loss_combine_fwd: null pointer");
  hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_cls, v_geo, v_sem1, v_sem2, wg, ws, out3);
  SUG_LAUNCH_CHECK("sug_loss_combine_fwd");
  return SUG_OK;
}

extern "C" int sug_loss_combine_bwd(const float* g, float wg, float ws, float* out4, void* stream) {
  SUG_REQUIRE(g && out4, "sug_loss_combine_bwd: null pointer");
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, wg, ws, out4);
  SUG_LAUNCH_CHECK("sug_loss_combine_bwd");
  return SUG_OK;
}
