// Per-point MLP layer fused with the max over a group of rows: the last 1x1 conv of PointNet's
// trunk and of its T-Nets followed by the global max over the N points (model/Model.py:274-279,
// model/model_utils.py:72-79, model/model_pointnet.py:36-48) and the last layer of a PointNet++
// set-abstraction MLP followed by the max over the nsample grouped points
// (model/pointnet2_utils.py:193-207).  The reference writes y = conv(x) ([B,1024,N] resp.
// [B,C,nsample,npoint]), normalises it (train-mode BatchNorm), applies ReLU and only then reduces;
// here y exists only as fp32 MFMA accumulator tiles.
//
// BatchNorm followed by (Leaky)ReLU is monotone per channel, so max_rows act(bn(y)) = act(bn(ext(y)))
// with ext = max for gamma >= 0 and min for gamma < 0: per (segment, channel) the kernel keeps the
// extreme pre-BN value and its row, and accumulates the BatchNorm sums of y on the way.
//
// Geometry: workgroup = 4 waves = (row block, 128 output channels); a wave owns 32 channels, whose
// weight rows stay in registers as the B operand (lane l: channel l&31, k = 2s + (l>>5));
// 32-row tiles of x stream global -> registers -> LDS (mfma_tile.h) and feed the A operand;
// y tile (rows x channels) = 16 accumulator registers per lane: channel l&31, rows
// (r&3) + 8*(r>>2) + 4*(l>>5).  The epilogue of tile t (bias, BN sums, running extreme) is a block of VALU
// work behind the MFMA chain of tile t+1 (the tile's LDS staging and global loads surround both).
//
// FLOPs 2*R*K*Co on the fp32 matrix pipe (157 TFLOP/s peak); algorithmic bytes
// 4*R*K (x, once per 128-channel block through L2) + 4*Co*K + 8*(R/L)*Co.
#include "common.h"
#include "mfma_tile.h"

namespace {
using namespace sug_tile;

// XF (round 5): the rows of x are PRE-activation outputs of the previous layer; its BatchNorm + (Leaky)ReLU -- per input
// channel scale / shift `xcoef` [2][CP] and slope `xslope` -- is applied to every tile on its way from the load registers to
// LDS, and (zout != nullptr, first channel block only) the activated rows are written out once for the backward.  The
// previous layer's separate BatchNorm pass (read y, write z) and this kernel's read of z collapse into one read of y.
template <int CP, bool XF = false>
__global__ __launch_bounds__(256, 2) void pointmlp_max_kernel(
    const float* __restrict__ x, int64_t ldx, int R, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma, int Co, int L, int rows_per_wg, int nrb,
    float* __restrict__ zext, int32_t* __restrict__ arg, float* __restrict__ ws, int pivoted, int parts,
    const float* __restrict__ xcoef = nullptr, float xslope = 0.f, float* __restrict__ zout = nullptr, int64_t ldz = 0) {
  // parts > 1 (small grids, pointmlp_max_fwd): a segment of L rows is split over `parts` workgroups of rows_per_wg = L / parts
  // rows; each writes its sign-folded partial extreme + row to the scratch behind the statistics rows of ws
  // ([nrb][Co] floats, then [nrb][Co] ints) and pointmlp_max_combine_kernel picks the first extreme in part order.
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);    // [3][TJ][RS]

  // (row block, channel block): the channel blocks of a row block re-read the same x rows, so they
  // get the same blockIdx % 8 (one XCD, one L2) when the row blocks split evenly over the XCDs
  const int ncb = Co >> 7;
  int rb, cb;
  if ((nrb & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    rb = (j / ncb) * 8 + xcd;
    cb = j % ncb;
  } else {
    rb = blockIdx.x / ncb;
    cb = blockIdx.x % ncb;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cj = lane & 31, h = lane >> 5;
  const int col = cb * 128 + wv * 32 + cj;
  const int row_begin = rb * rows_per_wg;
  const int nrows = (R - row_begin < rows_per_wg) ? R - row_begin : rows_per_wg;
  const float* xb = x + (int64_t)row_begin * ldx;

  // B operand: this lane's channel, k = 2s + h
  float bq[HALF];
  {
    const float* wr = W + (int64_t)col * CP;
#pragma unroll
    for (int g = 0; g < CP / 8; ++g) {
      const float4 lo = *reinterpret_cast<const float4*>(wr + 8 * g);
      const float4 hi = *reinterpret_cast<const float4*>(wr + 8 * g + 4);
      bq[4 * g + 0] = h ? lo.y : lo.x;
      bq[4 * g + 1] = h ? lo.w : lo.z;
      bq[4 * g + 2] = h ? hi.y : hi.x;
      bq[4 * g + 3] = h ? hi.w : hi.z;
    }
  }
  // XF: scale / shift of the 8 input features this thread stages (chunk c8 = threadIdx.x % (CP / 8) of every row it loads)
  float xsc[8], xsh[8];
  if constexpr (XF) {
    const int c8 = (int)threadIdx.x % (CP / 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xsc[e] = xcoef[c8 * 8 + e];
      xsh[e] = xcoef[CP + c8 * 8 + e];
    }
  }
  auto xf1 = [&](float v, float sc, float sh) {
    const float t = fmaf(v, sc, sh);
    return t > 0.f ? t : xslope * t;
  };
  // activate the tile a thread holds in registers (rows row0 + item / CH of this workgroup) and, for the first channel
  // block, write the activated rows out
  float* zb = (XF && zout != nullptr && cb == 0) ? zout + (int64_t)row_begin * ldz : nullptr;
  auto xform = [&](TileRegs<CP>& t, int row0, int nrows_) {
    if constexpr (XF) {
      constexpr int CH = CP / 8;
#pragma unroll
      for (int u = 0; u < TileRegs<CP>::NV; ++u) {
        float4 a = t.lo[u], b = t.hi[u];
        a.x = xf1(a.x, xsc[0], xsh[0]); a.y = xf1(a.y, xsc[1], xsh[1]); a.z = xf1(a.z, xsc[2], xsh[2]); a.w = xf1(a.w, xsc[3], xsh[3]);
        b.x = xf1(b.x, xsc[4], xsh[4]); b.y = xf1(b.y, xsc[5], xsh[5]); b.z = xf1(b.z, xsc[6], xsh[6]); b.w = xf1(b.w, xsc[7], xsh[7]);
        t.lo[u] = a;
        t.hi[u] = b;
        if (zb != nullptr) {
          const int item = (int)threadIdx.x + u * 256;
          const int r = row0 + item / CH;
          if (r < nrows_) {
            float* d = zb + (int64_t)r * ldz + (item % CH) * 8;
            *reinterpret_cast<float4*>(d) = a;
            *reinterpret_cast<float4*>(d + 4) = b;
          }
        }
      }
    }
  };
  const float sgn = gamma[col] >= 0.f ? 1.f : -1.f;
  // The sign of gamma is folded into the weights and the bias: the accumulator then holds sgn * y EXACTLY (negation commutes
  // with every rounding of the fma chain), the running extreme is a plain maximum, and the BatchNorm sums are taken of
  // sgn * (y - pivot): sum(sgn * v) = sgn * sum(v) and the squares do not see the sign, so s1 is un-folded once at the end.
#pragma unroll
  for (int e = 0; e < HALF; ++e) bq[e] *= sgn;
  const float bj = bias ? sgn * bias[col] : 0.f;
  // pivot of the BatchNorm sums (common.h): y of the launch's first row for this channel, the same fma chain in every
  // workgroup (each lane half over its half of k, halves added): the sums below are taken about it
  float pvt = 0.f;
  if (pivoted) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < HALF; ++e) {
      float xv = x[2 * e + h];
      if constexpr (XF) xv = xf1(xv, xcoef[2 * e + h], xcoef[CP + 2 * e + h]);
      d = fmaf(xv, bq[e], d);
    }
    pvt = __fadd_rn(d + __shfl_xor(d, 32), bj);          // sgn * (pivot of y)
    if (rb == 0 && h == 0) ws[SUG_PIVOT_OFFSET(Co) + col] = sgn * pvt;
  }

  float s1 = 0.f, s2 = 0.f;                          // BN sums of this lane's rows of this channel
  float best = -INFINITY;
  int barg = 0;
  const int ntile = (nrows + TJ - 1) / TJ;
  auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };

#define SUG_SB() __builtin_amdgcn_sched_barrier(0)
  // MFMA chain of the NEXT tile, back to back, then the epilogue of the CURRENT one as one block of VALU work.  (Rounds 2-3
  // issued the chain in pieces between the epilogue steps; on gfx950 fp32 MFMA and VALU do not overlap on a SIMD, so there
  // is nothing to hide -- measured equal, tools/ab_pointmlp.py: 162.0 vs 162.3 us at 64 x 1024 rows, K = 128, Co = 1024;
  // 204.7 vs 200.9 us at the sa1 shape -- and this form is the simpler one.)
  auto step = [&](const f32x16& acc_cur, f32x16& acc_next, const float* __restrict__ arow, int t) {
    // every tile is full: a workgroup's rows are whole segments (or equal parts of one) and L % 32 == 0
    const int segrow = (row_begin + t * TJ) % L + 4 * h;   // row of the tile's first row inside its segment
    // (the bias / pivot / BatchNorm-sum steps as packed fp32 operations -- v_pk_add_f32, v_pk_fma_f32 on register pairs --
    // measured 2 % SLOWER than the scalar forms below: tools/ab_pointmlp.py, 177.7 vs 173.7 us at the sa1 shape)
    auto epi = [&](int c) {
      const int rt = (c & 3) + 8 * (c >> 2);       // row inside the tile (+ 4h)
      const float y = __fadd_rn(acc_cur[c], bj);   // sgn * (x.w + b)
      const float yv = y - pvt;
      s1 += yv;
      s2 = fmaf(yv, yv, s2);
      const bool up = y > best;                    // strict: the first extreme of a lane's ascending rows wins
      best = up ? y : best;
      barg = up ? segrow + rt : barg;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_next[r] = 0.f;
    SUG_SB();
#pragma unroll
    for (int g = 0; g < HALF / 4; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq[4 * g + 0], acc_next, 0, 0, 0);
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq[4 * g + 1], acc_next, 0, 0, 0);
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq[4 * g + 2], acc_next, 0, 0, 0);
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq[4 * g + 3], acc_next, 0, 0, 0);
    }
    SUG_SB();
#pragma unroll
    for (int c = 0; c < 16; ++c) epi(c);
    SUG_SB();
  };

  // a segment ends with this tile: merge the two lane halves (other rows, same channel) and emit
  auto flush = [&](int t) {
    const float pb = __shfl_xor(best, 32);
    const int pa = __shfl_xor(barg, 32);
    const bool take = pb > best || (pb == best && pa < barg);
    const float fb = take ? pb : best;
    const int fa = take ? pa : barg;
    if (h == 0) {
      if (parts > 1) {
        float* pext = ws + (size_t)nrb * 2 * Co;
        int32_t* parg = reinterpret_cast<int32_t*>(pext + (size_t)nrb * Co);
        pext[(size_t)rb * Co + col] = fb;            // sign-folded: the combine kernel compares, then unfolds
        parg[(size_t)rb * Co + col] = fa;
      } else {
        const int64_t o = (int64_t)((row_begin + t * TJ) / L) * Co + col;
        zext[o] = sgn * fb;
        arg[o] = fa;
      }
    }
    best = -INFINITY;
    barg = 0;
  };

  {
    TileRegs<CP> tra, trb;
    tile_load<CP>(tra, xb, ldx, nrows, 0);
    xform(tra, 0, nrows);
    tile_store<CP, false>(tra, tbuf(0), nullptr, nrows, 0);
    __syncthreads();
    if (ntile > 1) tile_load<CP>(trb, xb, ldx, nrows, TJ);
    if (ntile > 2) tile_load<CP>(tra, xb, ldx, nrows, 2 * TJ);
    f32x16 acc_cur;
    {
      const float* arow = tbuf(0) + cj * RS + h * HALF;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_cur[r] = 0.f;
#pragma unroll
      for (int g = 0; g < HALF / 4; ++g) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq[4 * g + 0], acc_cur, 0, 0, 0);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq[4 * g + 1], acc_cur, 0, 0, 0);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq[4 * g + 2], acc_cur, 0, 0, 0);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq[4 * g + 3], acc_cur, 0, 0, 0);
      }
    }
    if (ntile > 1) {
      xform(trb, TJ, nrows);
      tile_store<CP, false>(trb, tbuf(1), nullptr, nrows, TJ);
    }
    __syncthreads();
    if (ntile > 3) tile_load<CP>(trb, xb, ldx, nrows, 3 * TJ);
    // iteration t: MFMA chain of tile t+1 || epilogue of tile t; registers of tile t+2 -> LDS; global
    // loads of tiles t+3, t+4 in flight (the last iteration's chain runs on a stale buffer, unused)
#define SUG_PM_BODY(T, TR) do { \
      f32x16 acc_next; \
      step(acc_cur, acc_next, tbuf((T) + 1) + cj * RS + h * HALF, (T)); \
      if ((((T) + 1) * TJ) % L == 0 || (T) + 1 == ntile) flush((T)); \
      if ((T) + 2 < ntile) { xform(TR, ((T) + 2) * TJ, nrows); tile_store<CP, false>(TR, tbuf((T) + 2), nullptr, nrows, ((T) + 2) * TJ); } \
      __syncthreads(); \
      if ((T) + 4 < ntile) tile_load<CP>(TR, xb, ldx, nrows, ((T) + 4) * TJ); \
      acc_cur = acc_next; \
    } while (0)
    for (int t = 0; t < ntile; t += 2) {
      SUG_PM_BODY(t, tra);
      if (t + 1 < ntile) SUG_PM_BODY(t + 1, trb);
    }
#undef SUG_PM_BODY
  }
#undef SUG_SB
  // BN partial sums of this (row block, channel): fixed order (lane half 0 + lane half 1)
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (h == 0) {
    ws[(size_t)rb * 2 * Co + col] = sgn * s1;
    ws[(size_t)rb * 2 * Co + Co + col] = s2;
  }
}

// Segments split over `parts` workgroups (small grids): the first extreme in ascending row order = the first part, in part
// order, whose sign-folded value is strictly larger than all earlier ones.
__global__ void pointmlp_max_combine_kernel(const float* __restrict__ ws, const float* __restrict__ gamma, int Co, int S,
                                            int parts, int nrb, float* __restrict__ zext, int32_t* __restrict__ arg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * Co) return;
  const int s = (int)(i / Co), col = (int)(i % Co);
  const float* pext = ws + (size_t)nrb * 2 * Co;
  const int32_t* parg = reinterpret_cast<const int32_t*>(pext + (size_t)nrb * Co);
  float best = -INFINITY;
  int barg = 0;
  for (int p = 0; p < parts; ++p) {
    const float v = pext[(size_t)(s * parts + p) * Co + col];
    const int a = parg[(size_t)(s * parts + p) * Co + col];
    if (v > best) { best = v; barg = a; }
  }
  zext[i] = (gamma[col] >= 0.f ? 1.f : -1.f) * best;
  arg[i] = barg;
}

// ---------------------------------------------------------------------------------------------
// Plain per-point linear layer on the same pipeline: y[r, :] = x[r, :] . w^T (+ bias), y stored.
// Used for the EdgeConv operand PQ = x . [W1 ; W2-W1]^T (model_utils.py:188-210 after the algebra of
// edgeconv.hip) at K = 64 / 128, where the library's fp32 GEMM selection runs at 30-60 TFLOP/s.
// Same geometry as pointmlp_max_kernel; the y tile of iteration t is stored while the MFMA chain of
// tile t+1 runs: register r of a lane = row (r&3) + 8*(r>>2) + 4*(l>>5), 32 consecutive channels per
// half-wave = one 128-byte segment per row.
template <int CP>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(const float* __restrict__ x, int64_t ldx, int R,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           int Co, int rows_per_wg, int nrb, float* __restrict__ y,
                                                           int64_t ldy) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);    // [3][TJ][RS]
  const int ncb = Co >> 7;
  int rb, cb;
  if ((nrb & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    rb = (j / ncb) * 8 + xcd;
    cb = j % ncb;
  } else {
    rb = blockIdx.x / ncb;
    cb = blockIdx.x % ncb;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cj = lane & 31, h = lane >> 5;
  const int col = cb * 128 + wv * 32 + cj;
  const int row_begin = rb * rows_per_wg;
  const int nrows = (R - row_begin < rows_per_wg) ? R - row_begin : rows_per_wg;
  const float* xb = x + (int64_t)row_begin * ldx;
  float bq[HALF];
  {
    const float* wr = W + (int64_t)col * CP;
#pragma unroll
    for (int g = 0; g < CP / 8; ++g) {
      const float4 lo = *reinterpret_cast<const float4*>(wr + 8 * g);
      const float4 hi = *reinterpret_cast<const float4*>(wr + 8 * g + 4);
      bq[4 * g + 0] = h ? lo.y : lo.x;
      bq[4 * g + 1] = h ? lo.w : lo.z;
      bq[4 * g + 2] = h ? hi.y : hi.x;
      bq[4 * g + 3] = h ? hi.w : hi.z;
    }
  }
  const float bj = bias ? bias[col] : 0.f;
  const int ntile = (nrows + TJ - 1) / TJ;
  auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };
  auto chain = [&](const float* __restrict__ arow) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < HALF / 4; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq[4 * g + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq[4 * g + 3], acc, 0, 0, 0);
    }
    return acc;
  };
  auto store = [&](const f32x16& acc, int t) {
    float* yb = y + (int64_t)(row_begin + t * TJ + 4 * h) * ldy + col;
    const int lim = nrows - t * TJ - 4 * h;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int rt = (c & 3) + 8 * (c >> 2);
      if (rt < lim) yb[(int64_t)rt * ldy] = __fadd_rn(acc[c], bj);
    }
  };
  TileRegs<CP> tra, trb;
  tile_load<CP>(tra, xb, ldx, nrows, 0);
  tile_store<CP, false>(tra, tbuf(0), nullptr, nrows, 0);
  __syncthreads();
  if (ntile > 1) tile_load<CP>(trb, xb, ldx, nrows, TJ);
  if (ntile > 2) tile_load<CP>(tra, xb, ldx, nrows, 2 * TJ);
  f32x16 acc_cur = chain(tbuf(0) + cj * RS + h * HALF);
  if (ntile > 1) tile_store<CP, false>(trb, tbuf(1), nullptr, nrows, TJ);
  __syncthreads();
  if (ntile > 3) tile_load<CP>(trb, xb, ldx, nrows, 3 * TJ);
#define SUG_RG_BODY(T, TR) do { \
    f32x16 acc_next = chain(tbuf((T) + 1) + cj * RS + h * HALF); \
    store(acc_cur, (T)); \
    if ((T) + 2 < ntile) tile_store<CP, false>(TR, tbuf((T) + 2), nullptr, nrows, ((T) + 2) * TJ); \
    __syncthreads(); \
    if ((T) + 4 < ntile) tile_load<CP>(TR, xb, ldx, nrows, ((T) + 4) * TJ); \
    acc_cur = acc_next; \
  } while (0)
  for (int t = 0; t < ntile; t += 2) {
    SUG_RG_BODY(t, tra);
    if (t + 1 < ntile) SUG_RG_BODY(t + 1, trb);
  }
#undef SUG_RG_BODY
}

// ---------------------------------------------------------------------------------------------
// Backward, the part that follows the arg-extreme rows.  With a[s,c] = scale_c * gout * act'
// (sug_edgeconv_bwd_reduce on the [S,Co] tensors) and n*(s,c) = s*L + arg[s,c]:
//     dx[n*(s,c), :] += a[s,c] * W[c, :]            (rows of x that won a channel)
//     dW[c, :]       += a[s,c] * x[n*(s,c), :]
// The BatchNorm-statistics terms of the gradient are dense in the rows but of rank K: the caller adds
// them as -(x.A + v) and a [Co,K] matrix built from X^T X (see ops.PointMLPMax).
//
// dx: one workgroup per (segment, chunk of 128 rows); a wave owns rows, walks the channels that
// picked each row in ascending channel order (ballot), so the sum order is fixed.
template <int KV>     // K / 64 floats per lane
__global__ __launch_bounds__(256) void pointmlp_bwd_dx_kernel(const float* __restrict__ a, const int32_t* __restrict__ arg,
                                                              const float* __restrict__ W, int Co, int L, int S,
                                                              int chunks, float* __restrict__ dx, int64_t lddx) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float* s_a = s_mem;                                        // [Co]
  int* s_arg = reinterpret_cast<int*>(s_mem + Co);           // [Co]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int K = KV * 64;
  const int rows_per_chunk = (L + chunks - 1) / chunks;
  for (int item = blockIdx.x; item < S * chunks; item += gridDim.x) {
    const int s = item / chunks, ch = item % chunks;
    __syncthreads();
    for (int c = threadIdx.x; c < Co; c += 256) {
      s_a[c] = a[(int64_t)s * Co + c];
      s_arg[c] = arg[(int64_t)s * Co + c];
    }
    __syncthreads();
    const int r0 = ch * rows_per_chunk;
    const int r1 = (r0 + rows_per_chunk < L) ? r0 + rows_per_chunk : L;
    if (Co <= 256) {
      // few channels per row: every row scans all of them (a ballot per 64 channels)
      for (int r = r0 + wv; r < r1; r += 4) {
        float acc[KV];
#pragma unroll
        for (int u = 0; u < KV; ++u) acc[u] = 0.f;
        bool any = false;
        for (int c0 = 0; c0 < Co; c0 += 64) {
          unsigned long long m = __ballot(c0 + lane < Co && s_arg[c0 + lane] == r);
          while (m) {
            const int c = c0 + __builtin_ctzll(m);
            m &= m - 1;
            const float av = s_a[c];
            const float* wr = W + (int64_t)c * K + lane * KV;
#pragma unroll
            for (int u = 0; u < KV; ++u) acc[u] = fmaf(av, wr[u], acc[u]);
            any = true;
          }
        }
        if (any) {
          float* d = dx + ((int64_t)s * L + r) * lddx + lane * KV;
#pragma unroll
          for (int u = 0; u < KV; ++u) d[u] += acc[u];
        }
      }
      continue;
    }
    // Many channels per row block (the global-max layers: every row scanned all Co channels for the ~1 that picked
    // it): a wave owns a contiguous quarter of the chunk's rows, compacts -- in ascending channel order, one ballot
    // per 64 channels -- the channels whose arg falls into its quarter, and every row scans only that short list.
    // Same channels in the same order per row: the sums are bit for bit those of the full scan.
    const int per_wave = (r1 - r0 + 3) / 4;
    const int ra = r0 + wv * per_wave, rb = (ra + per_wave < r1) ? ra + per_wave : r1;
    int* s_list = reinterpret_cast<int*>(s_mem + 2 * Co) + wv * Co;            // [4][Co] compacted channels
    int nlist = 0;
    for (int c0 = 0; c0 < Co; c0 += 64) {
      const int c = c0 + lane;
      const bool in = c < Co && s_arg[c] >= ra && s_arg[c] < rb;
      const unsigned long long m = __ballot(in);
      if (in) s_list[nlist + __popcll(m & ((1ull << lane) - 1ull))] = c;
      nlist += __popcll(m);
    }
    for (int r = ra; r < rb; ++r) {
      float acc[KV];
#pragma unroll
      for (int u = 0; u < KV; ++u) acc[u] = 0.f;
      bool any = false;
      for (int i0 = 0; i0 < nlist; i0 += 64) {
        const int cl = (i0 + lane < nlist) ? s_list[i0 + lane] : -1;
        unsigned long long m = __ballot(cl >= 0 && s_arg[cl >= 0 ? cl : 0] == r);
        while (m) {
          const int c = s_list[i0 + __builtin_ctzll(m)];
          m &= m - 1;
          const float av = s_a[c];
          const float* wr = W + (int64_t)c * K + lane * KV;
#pragma unroll
          for (int u = 0; u < KV; ++u) acc[u] = fmaf(av, wr[u], acc[u]);
          any = true;
        }
      }
      if (any) {
        float* d = dx + ((int64_t)s * L + r) * lddx + lane * KV;
#pragma unroll
        for (int u = 0; u < KV; ++u) d[u] += acc[u];
      }
    }
  }
}

// The same for short segments (L <= 64 rows, Co <= 256: the set-abstraction layers, 65 k segments of 32 rows at
// config 3): one WAVE per segment, its a / arg values in registers (channel = lane + 64 i), no LDS staging and
// no workgroup barriers (the kernel above spends its time in two barriers per 32-row item).  Channels of a row
// are taken in ascending order (i, then lane): the sums are those of the kernel above.
template <int KV, int CV>
__global__ __launch_bounds__(256) void pointmlp_bwd_dx_seg_kernel(const float* __restrict__ a, const int32_t* __restrict__ arg,
                                                                  const float* __restrict__ W, int Co, int L, int S,
                                                                  float* __restrict__ dx, int64_t lddx) {
  constexpr int K = KV * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int s = blockIdx.x * 4 + wv; s < S; s += gridDim.x * 4) {
    float av[CV];
    int rg[CV];
#pragma unroll
    for (int i = 0; i < CV; ++i) {
      const int c = lane + 64 * i;
      av[i] = c < Co ? a[(int64_t)s * Co + c] : 0.f;
      rg[i] = c < Co ? arg[(int64_t)s * Co + c] : -1;
    }
    for (int r = 0; r < L; ++r) {
      float acc[KV];
#pragma unroll
      for (int u = 0; u < KV; ++u) acc[u] = 0.f;
      bool any = false;
#pragma unroll
      for (int i = 0; i < CV; ++i) {
        unsigned long long m = __ballot(rg[i] == r);
        while (m) {
          // up to NB channels per trip: their weight rows are requested together (one dependent load per
          // channel left the wave waiting on L2 latency ~130 times per segment); absent slots repeat the first
          // channel with a zero coefficient, the order of the sums is unchanged
          constexpr int NB = CV <= 2 ? 4 : 1;     // (Co = 256 over 64 rows: one channel per ballot on average)
          int l[NB];
          float ac[NB];
          l[0] = __builtin_ctzll(m);
          m &= m - 1;
#pragma unroll
          for (int t = 1; t < NB; ++t) {
            l[t] = m ? (int)__builtin_ctzll(m) : -1;
            if (m) m &= m - 1;
          }
          float wv4[NB][KV];
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            const int lt = l[t] >= 0 ? l[t] : l[0];
            ac[t] = l[t] >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av[i]), lt)) : 0.f;
            const float* wr = W + (int64_t)(lt + 64 * i) * K + lane * KV;
#pragma unroll
            for (int u = 0; u < KV; ++u) wv4[t][u] = wr[u];
          }
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            if (l[t] >= 0) {
#pragma unroll
              for (int u = 0; u < KV; ++u) acc[u] = fmaf(ac[t], wv4[t][u], acc[u]);
            }
          }
          any = true;
        }
      }
      if (any) {
        float* d = dx + ((int64_t)s * L + r) * lddx + lane * KV;
#pragma unroll
        for (int u = 0; u < KV; ++u) d[u] += acc[u];
      }
    }
  }
}

// dW partials: grid (channel blocks of 64, segment chunks); a wave owns 16 channels and sums
// a[s,c] * x[n*(s,c),:] over the segments of its chunk in ascending s; one partial row per chunk
// (dwp [nsc][Co][K]), folded in order by pointmlp_bwd_dw_fold_kernel.
template <int KV>
__global__ __launch_bounds__(256) void pointmlp_bwd_dw_kernel(const float* __restrict__ a, const int32_t* __restrict__ arg,
                                                              const float* __restrict__ x, int64_t ldx, int Co, int L,
                                                              int S, int nsc, float* __restrict__ dwp) {
  constexpr int K = KV * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64 + wv * 16;
  const int sc = blockIdx.y;
  const int per = (S + nsc - 1) / nsc;
  const int sa = sc * per, sb = (sa + per < S) ? sa + per : S;
  float acc[16][KV];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int u = 0; u < KV; ++u) acc[i][u] = 0.f;
  const bool mine = lane < 16 && c0 + lane < Co;
  float av_n = 0.f;
  int rw_n = 0;
  if (mine && sa < sb) {
    av_n = a[(int64_t)sa * Co + c0 + lane];
    rw_n = arg[(int64_t)sa * Co + c0 + lane];
  }
  for (int s = sa; s < sb; ++s) {
    const float av = av_n;
    const int rw = rw_n;
    if (mine && s + 1 < sb) {                    // the next segment's coefficients fly under this one's gathers
      av_n = a[(int64_t)(s + 1) * Co + c0 + lane];
      rw_n = arg[(int64_t)(s + 1) * Co + c0 + lane];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float ai = __shfl(av, i);
      const int ri = __shfl(rw, i);
      const float* xr = x + ((int64_t)s * L + ri) * ldx + lane * KV;
#pragma unroll
      for (int u = 0; u < KV; ++u) acc[i][u] = fmaf(ai, xr[u], acc[i][u]);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (c0 + i < Co) {
      float* d = dwp + ((int64_t)sc * Co + c0 + i) * K + lane * KV;
#pragma unroll
      for (int u = 0; u < KV; ++u) d[u] = acc[i][u];
    }
  }
}

__global__ __launch_bounds__(256) void pointmlp_bwd_dw_fold_kernel(const float* __restrict__ dwp, int nsc, int64_t n,
                                                                   float* __restrict__ dw) {
  // 16 elements x 16 chunk-lanes per workgroup (chunk-lane p sums chunks p, p+16, ... in fp64), combined in lane order
  __shared__ double s_p[16][17];
  const int el = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int64_t e = (int64_t)blockIdx.x * 16 + el;
  double t = 0.0;
  if (e < n) {
    int i = p;
    for (; i + 112 < nsc; i += 128) {             // eight partial rows in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dwp[(int64_t)(i + 16 * u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (double)v[u];
    }
    for (; i < nsc; i += 16) t += (double)dwp[(int64_t)i * n + e];
  }
  s_p[p][el] = t;
  __syncthreads();
  if (p == 0 && e < n) {
    double r = s_p[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) r += s_p[i][el];
    dw[e] = (float)r;
  }
}

// The dense BatchNorm-statistics terms of the backward are of rank K: dy = a_full - k1 - k2*y with y = x.W^T + b
// gives dx = ... - (x.A + v), A = W^T diag(k2) W, v = (k1 + k2*b).W, and dW = ... - kb (x) sum(x) - diag(k2) W.(X^T X).
// Coefficients from the folded statistics (fp64, as the torch expressions they replace):
//   k2 = scale/M * rstd * dgamma,  kb = scale/M * (dbeta - mean*rstd*dgamma) + k2*b
// One elementwise launch writes kb, k2, -kb and -diag(k2).W; the two small products -A = (-diag(k2) W)^T.W and
// -v = (-kb).W are left to the library (a one-workgroup-per-row version of them ran at 94 us: 1024 dependent steps).
__global__ __launch_bounds__(256) void pointmlp_bwd_coef_kernel(const float* __restrict__ coef, const double* __restrict__ red,
                                                                 const float* __restrict__ bias, const float* __restrict__ w,
                                                                 double M, int K, int Co, float* __restrict__ kb,
                                                                 float* __restrict__ k2, float* __restrict__ nkb,
                                                                 float* __restrict__ nwk) {
  const int c = blockIdx.x;
  const double scale = coef[c], mean = coef[2 * Co + c], rstd = coef[3 * Co + c];
  const double dbeta = red[c], dgamma = red[Co + c];
  const double k2d = scale / M * rstd * dgamma;
  const double k1d = scale / M * (dbeta - mean * rstd * dgamma);
  const float k2f = (float)k2d, kbf = (float)(k1d + (bias ? k2d * (double)bias[c] : 0.0));
  if (threadIdx.x == 0) {
    kb[c] = kbf;
    k2[c] = k2f;
    nkb[c] = -kbf;
  }
  for (int j = threadIdx.x; j < K; j += 256) nwk[(size_t)c * K + j] = -(k2f * w[(size_t)c * K + j]);
}

// dw[c,:] += dws[c,:] - (kb[c]*sx[:] + k2[c] * w[c,:].XtX)   (with_stats = 0: dw += dws)
__global__ __launch_bounds__(128) void pointmlp_bwd_dwfix_kernel(float* __restrict__ dw, const float* __restrict__ dws,
                                                                  const float* __restrict__ kb, const float* __restrict__ k2,
                                                                  const float* __restrict__ w, const float* __restrict__ xtx,
                                                                  const float* __restrict__ sx, int K, int Co, int with_stats) {
  __shared__ float s_w[128];
  const int c = blockIdx.x, j = threadIdx.x;
  if (j < K) s_w[j] = w[(size_t)c * K + j];
  __syncthreads();
  if (j >= K) return;
  float corr = 0.f;
  if (with_stats) {
    float acc = 0.f;
    for (int i = 0; i < K; ++i) acc = fmaf(s_w[i], xtx[(size_t)i * K + j], acc);
    corr = kb[c] * sx[j] + k2[c] * acc;
  }
  dw[(size_t)c * K + j] += dws[(size_t)c * K + j] - corr;
}

}  // namespace

static int pointmlp_rows_per_wg(int64_t rows, int L) {
  // whole segments per workgroup, ~1024 rows, at most SUG_STATS_BLOCKS row blocks
  int64_t rpw = L >= 1024 ? L : (1024 / L) * L;
  while ((rows + rpw - 1) / rpw > SUG_STATS_ROWS) rpw *= 2;
  return (int)rpw;
}

static int pointmlp_max_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias,
                            const float* gamma, int Co, int seg, float* zext, int32_t* arg, float* ws, int* nblk,
                            void* stream, int pivoted, const float* xcoef, float xslope, float* zout, int64_t ldz);

extern "C" int sug_pointmlp_max_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias,
                                    const float* gamma, int Co, int seg, float* zext, int32_t* arg, float* ws,
                                    int* nblk, void* stream) {
  return pointmlp_max_fwd(x, ldx, rows, K, w, bias, gamma, Co, seg, zext, arg, ws, nblk, stream, 0, nullptr, 0.f, nullptr, 0);   // plain sums
}

static int pointmlp_max_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias,
                            const float* gamma, int Co, int seg, float* zext, int32_t* arg, float* ws, int* nblk,
                            void* stream, int pivoted, const float* xcoef, float xslope, float* zout, int64_t ldz) {
  SUG_REQUIRE(x && w && gamma && zext && arg && ws && nblk, "sug_pointmlp_max_fwd: null pointer");
  SUG_REQUIRE(K == 64 || K == 128, "sug_pointmlp_max_fwd: K=%d (64 or 128 input channels)", K);
  SUG_REQUIRE(Co > 0 && Co % 128 == 0, "sug_pointmlp_max_fwd: Co=%d must be a multiple of 128", Co);
  SUG_REQUIRE(seg >= 32 && seg % 32 == 0, "sug_pointmlp_max_fwd: segment length %d must be a multiple of 32", seg);
  SUG_REQUIRE(rows > 0 && rows % seg == 0 && rows < (1ll << 31), "sug_pointmlp_max_fwd: %lld rows are not whole segments of %d",
              (long long)rows, seg);
  SUG_REQUIRE(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0,
              "sug_pointmlp_max_fwd: x / w must be 16-byte aligned with row strides in multiples of 4");
  int rpw = pointmlp_rows_per_wg(rows, seg);
  int nrb = sug_divup(rows, rpw);
  // Small batches (config 1: 16 clouds of 1024 points x 8 channel blocks = 128 workgroups on 256 CUs, one wave per SIMD on
  // half of the chip): split every segment over `parts` workgroups so that the grid reaches two workgroups per CU; the
  // partial extremes go through the scratch behind the statistics rows and one small combine launch.  Same results (the
  // first extreme in ascending row order wins either way); the BatchNorm partial rows just become more.
  int parts = 1;
  static const int no_split = getenv("SUG_POINTMLP_NOSPLIT") ? atoi(getenv("SUG_POINTMLP_NOSPLIT")) : 0;
  if (rpw == seg && !no_split) {
    const int S = (int)(rows / seg), want = 2 * sug_cu_count();
    while (parts < 8 && (int64_t)S * parts * (Co / 128) < want && seg % (parts * 2 * 128) == 0 && 2 * S * parts * 2 <= SUG_STATS_ROWS)
      parts *= 2;
    if (parts > 1) {
      rpw = seg / parts;
      nrb = S * parts;
    }
  }
  const int grid = nrb * (Co / 128);
  hipStream_t st = (hipStream_t)stream;
  if (xcoef) {
    SUG_REQUIRE(!zout || (ldz >= K && ldz % 4 == 0 && ((uintptr_t)zout % 16) == 0), "sug_pointmlp_max_fwd: bad zout / ldz");
    if (K == 128)
      hipLaunchKernelGGL((pointmlp_max_kernel<128, true>), dim3(grid), dim3(256), (size_t)3 * TJ * (128 + 4) * sizeof(float), st, x,
                         ldx, (int)rows, w, bias, gamma, Co, seg, rpw, nrb, zext, arg, ws, pivoted, parts, xcoef, xslope, zout, ldz);
    else
      hipLaunchKernelGGL((pointmlp_max_kernel<64, true>), dim3(grid), dim3(256), (size_t)3 * TJ * (64 + 4) * sizeof(float), st, x,
                         ldx, (int)rows, w, bias, gamma, Co, seg, rpw, nrb, zext, arg, ws, pivoted, parts, xcoef, xslope, zout, ldz);
  } else if (K == 128) {
    const size_t sh = (size_t)3 * TJ * (128 + 4) * sizeof(float);
    hipLaunchKernelGGL((pointmlp_max_kernel<128>), dim3(grid), dim3(256), sh, st, x, ldx, (int)rows, w, bias, gamma, Co,
                       seg, rpw, nrb, zext, arg, ws, pivoted, parts, nullptr, 0.f, nullptr, (int64_t)0);
  } else {
    const size_t sh = (size_t)3 * TJ * (64 + 4) * sizeof(float);
    hipLaunchKernelGGL((pointmlp_max_kernel<64>), dim3(grid), dim3(256), sh, st, x, ldx, (int)rows, w, bias, gamma, Co,
                       seg, rpw, nrb, zext, arg, ws, pivoted, parts, nullptr, 0.f, nullptr, (int64_t)0);
  }
  SUG_LAUNCH_CHECK("sug_pointmlp_max_fwd");
  if (parts > 1) {
    const int64_t n = rows / seg * Co;
    hipLaunchKernelGGL(pointmlp_max_combine_kernel, dim3((unsigned)sug_divup(n, 256)), dim3(256), 0, st, ws, gamma, Co,
                       (int)(rows / seg), parts, nrb, zext, arg);
    SUG_LAUNCH_CHECK("sug_pointmlp_max_fwd(combine)");
  }
  *nblk = nrb;
  return SUG_OK;
}

extern "C" int sug_rows_gemm(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias, int Co,
                             float* y, int64_t ldy, void* stream) {
  SUG_REQUIRE(x && w && y, "sug_rows_gemm: null pointer");
  SUG_REQUIRE(K == 64 || K == 128, "sug_rows_gemm: K=%d (64 or 128 input channels)", K);
  SUG_REQUIRE(Co > 0 && Co % 128 == 0 && ldy >= Co, "sug_rows_gemm: Co=%d must be a multiple of 128 (ldy >= Co)", Co);
  SUG_REQUIRE(rows > 0 && rows < (1ll << 31), "sug_rows_gemm: bad row count");
  SUG_REQUIRE(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0,
              "sug_rows_gemm: x / w must be 16-byte aligned with row strides in multiples of 4");
  const int rpw = 1024;
  const int nrb = sug_divup(rows, rpw);
  const int grid = nrb * (Co / 128);
  hipStream_t st = (hipStream_t)stream;
  if (K == 128)
    hipLaunchKernelGGL((rows_gemm_kernel<128>), dim3(grid), dim3(256), (size_t)3 * TJ * (128 + 4) * sizeof(float), st, x, ldx,
                       (int)rows, w, bias, Co, rpw, nrb, y, ldy);
  else
    hipLaunchKernelGGL((rows_gemm_kernel<64>), dim3(grid), dim3(256), (size_t)3 * TJ * (64 + 4) * sizeof(float), st, x, ldx,
                       (int)rows, w, bias, Co, rpw, nrb, y, ldy);
  SUG_LAUNCH_CHECK("sug_rows_gemm");
  return SUG_OK;
}

static int pointmlp_max_layer(const float* x, int64_t ldx, int64_t rows, int K, const float* w,
                              const float* bias, const float* gamma, const float* beta, int Co, int seg,
                              int groups, int training, float eps, float momentum, float slope,
                              float* running_mean, float* running_var, float* zext, int32_t* arg,
                              float* coef, float* out, int64_t ldo, float* ws, void* stream, const float* xcoef, float xslope,
                              float* zout, int64_t ldz);

extern "C" int sug_pointmlp_max_layer_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w,
                                          const float* bias, const float* gamma, const float* beta, int Co, int seg,
                                          int groups, int training, float eps, float momentum, float slope,
                                          float* running_mean, float* running_var, float* zext, int32_t* arg,
                                          float* coef, float* out, int64_t ldo, float* ws, void* stream) {
  return pointmlp_max_layer(x, ldx, rows, K, w, bias, gamma, beta, Co, seg, groups, training, eps, momentum, slope, running_mean,
                            running_var, zext, arg, coef, out, ldo, ws, stream, nullptr, 0.f, nullptr, 0);
}

extern "C" int sug_pointmlp_max_layer_fwd_xf(const float* y, int64_t ldy, int64_t rows, int K, const float* xcoef, float xslope,
                                             float* zout, int64_t ldz, const float* w, const float* bias, const float* gamma,
                                             const float* beta, int Co, int seg, int groups, int training, float eps,
                                             float momentum, float slope, float* running_mean, float* running_var,
                                             float* zext, int32_t* arg, float* coef, float* out, int64_t ldo, float* ws,
                                             void* stream) {
  SUG_REQUIRE(xcoef, "sug_pointmlp_max_layer_fwd_xf: null input coefficients");
  return pointmlp_max_layer(y, ldy, rows, K, w, bias, gamma, beta, Co, seg, groups, training, eps, momentum, slope, running_mean,
                            running_var, zext, arg, coef, out, ldo, ws, stream, xcoef, xslope, zout, ldz);
}

static int pointmlp_max_layer(const float* x, int64_t ldx, int64_t rows, int K, const float* w,
                              const float* bias, const float* gamma, const float* beta, int Co, int seg,
                              int groups, int training, float eps, float momentum, float slope,
                              float* running_mean, float* running_var, float* zext, int32_t* arg,
                              float* coef, float* out, int64_t ldo, float* ws, void* stream, const float* xcoef, float xslope,
                              float* zout, int64_t ldz) {
  SUG_REQUIRE(groups >= 1 && rows > 0 && rows % ((int64_t)groups * seg) == 0,
              "sug_pointmlp_max_layer_fwd: %lld rows do not split into %d groups of whole segments", (long long)rows, groups);
  SUG_REQUIRE(beta && coef && out && ldo >= Co, "sug_pointmlp_max_layer_fwd: null pointer / bad ldo");
  const int64_t rg = rows / groups, sg = rg / seg;
  for (int g = 0; g < groups; ++g) {
    float* cg = coef + (int64_t)g * 5 * Co;
    int nblk = 0;
    // (input transform: group g's scale | shift rows of the previous layer's coefficient block [groups][5][K])
    int rc = pointmlp_max_fwd(x + g * rg * ldx, ldx, rg, K, w, bias, gamma, Co, seg, zext + g * sg * Co,
                              arg + g * sg * Co, ws, &nblk, stream, 1, xcoef ? xcoef + (int64_t)g * 5 * K : nullptr, xslope,
                              zout ? zout + g * rg * ldz : nullptr, ldz);
    if (rc != SUG_OK) return rc;
    if (training) {
      rc = sug_stats_finalize(ws, nblk, Co, gamma, beta, (double)rg, eps, momentum, running_mean, running_var, cg,
                              (hipStream_t)stream, ws + SUG_PIVOT_OFFSET(Co));
      if (rc != SUG_OK) return rc;
    }
    rc = sug_affine_act(zext + g * sg * Co, Co, cg, sg, Co, slope, out + g * sg * ldo, ldo, stream);
    if (rc != SUG_OK) return rc;
  }
  return SUG_OK;
}

extern "C" int sug_pointmlp_max_bwd_coef(const float* coef, const double* red, const float* bias, const float* w,
                                        int64_t rows, int K, int Co, float* kb, float* k2, float* nkb, float* nwk,
                                        void* stream) {
  SUG_REQUIRE(coef && red && w && kb && k2 && nkb && nwk, "sug_pointmlp_max_bwd_coef: null pointer");
  SUG_REQUIRE(K > 0 && Co > 0 && Co <= 65535 && rows > 0, "sug_pointmlp_max_bwd_coef: bad shape");
  hipLaunchKernelGGL(pointmlp_bwd_coef_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, coef, red, bias, w, (double)rows, K,
                     Co, kb, k2, nkb, nwk);
  SUG_LAUNCH_CHECK("sug_pointmlp_max_bwd_coef");
  return SUG_OK;
}

extern "C" int sug_pointmlp_max_bwd_dwfix(float* dw, const float* dws, const float* kb, const float* k2, const float* w,
                                         const float* xtx, const float* sx, int K, int Co, int with_stats, void* stream) {
  SUG_REQUIRE(dw && dws && w, "sug_pointmlp_max_bwd_dwfix: null pointer");
  SUG_REQUIRE(!with_stats || (kb && k2 && xtx && sx), "sug_pointmlp_max_bwd_dwfix: null pointer");
  SUG_REQUIRE(K > 0 && K <= 128 && Co > 0, "sug_pointmlp_max_bwd_dwfix: bad shape");
  hipLaunchKernelGGL(pointmlp_bwd_dwfix_kernel, dim3(Co), dim3(128), 0, (hipStream_t)stream, dw, dws, kb, k2, w, xtx, sx, K,
                     Co, with_stats);
  SUG_LAUNCH_CHECK("sug_pointmlp_max_bwd_dwfix");
  return SUG_OK;
}

// segment chunks of the dW pass: every wave walks its chunk serially (a / arg of a segment, then 16 gathered rows:
// one L2 latency per segment), so many short chunks hide that latency by occupancy; 1024 chunks of >= 16 segments
static int64_t dw_chunks(int64_t S) {
  int64_t nsc = S / 16;
  if (nsc > 1024) nsc = 1024;
  if (nsc < 256) nsc = S < 256 ? S : 256;
  return nsc < 1 ? 1 : nsc;
}

extern "C" int64_t sug_pointmlp_max_bwd_workspace(int64_t rows, int K, int Co, int seg) {
  const int64_t S = rows / (seg > 0 ? seg : 1);
  return dw_chunks(S) * (int64_t)Co * K;
}

extern "C" int sug_pointmlp_max_bwd_sparse(const float* a, const int32_t* arg, const float* x, int64_t ldx,
                                           const float* w, int64_t rows, int K, int Co, int seg, float* dx,
                                           int64_t lddx, float* dw, float* ws, void* stream) {
  SUG_REQUIRE(a && arg && x && w && dx && dw && ws, "sug_pointmlp_max_bwd_sparse: null pointer");
  SUG_REQUIRE(K == 64 || K == 128, "sug_pointmlp_max_bwd_sparse: K=%d (64 or 128)", K);
  SUG_REQUIRE(Co > 0 && Co <= 4096 && seg > 0 && rows > 0 && rows % seg == 0 && ldx >= K && lddx >= K,
              "sug_pointmlp_max_bwd_sparse: bad shape");
  const int64_t S = rows / seg;
  SUG_REQUIRE(S < (1ll << 30), "sug_pointmlp_max_bwd_sparse: too many segments");
  hipStream_t st = (hipStream_t)stream;
  // rows per workgroup of the dx pass: 128, or fewer (down to 16) when that grid would leave CUs idle (config 1: 8 segments
  // of 1024 rows per domain group = 64 workgroups; the sums of a row do not depend on the chunking)
  int chunks = sug_divup(seg, 128);
  {
    const int64_t want = 2 * (int64_t)sug_cu_count();
    while (S * chunks < want && seg / (chunks * 2) >= 16) chunks *= 2;
  }
  int64_t g64 = S * chunks;
  const int grid = (int)(g64 < 4096 ? g64 : 4096);
  const size_t sh = (size_t)Co * 8 + (Co > 256 ? (size_t)Co * 16 : 0);      // a, arg (+ 4 compacted channel lists)
  const int nsc = (int)dw_chunks(S);
  dim3 g2(sug_divup(Co, 64), nsc);
  const bool per_wave = seg <= 64 && Co <= 256;          // short segments: a wave per segment, no barriers
  const int gseg = (int)(S / 4 < 8192 ? (S + 3) / 4 : 8192);
  if (K == 128) {
    if (per_wave && Co <= 128)
      hipLaunchKernelGGL((pointmlp_bwd_dx_seg_kernel<2, 2>), dim3(gseg), dim3(256), 0, st, a, arg, w, Co, seg, (int)S, dx, lddx);
    else if (per_wave)
      hipLaunchKernelGGL((pointmlp_bwd_dx_seg_kernel<2, 4>), dim3(gseg), dim3(256), 0, st, a, arg, w, Co, seg, (int)S, dx, lddx);
    else
      hipLaunchKernelGGL((pointmlp_bwd_dx_kernel<2>), dim3(grid), dim3(256), sh, st, a, arg, w, Co, seg, (int)S, chunks, dx, lddx);
    hipLaunchKernelGGL((pointmlp_bwd_dw_kernel<2>), g2, dim3(256), 0, st, a, arg, x, ldx, Co, seg, (int)S, nsc, ws);
  } else {
    if (per_wave && Co <= 128)
      hipLaunchKernelGGL((pointmlp_bwd_dx_seg_kernel<1, 2>), dim3(gseg), dim3(256), 0, st, a, arg, w, Co, seg, (int)S, dx, lddx);
    else if (per_wave)
      hipLaunchKernelGGL((pointmlp_bwd_dx_seg_kernel<1, 4>), dim3(gseg), dim3(256), 0, st, a, arg, w, Co, seg, (int)S, dx, lddx);
    else
      hipLaunchKernelGGL((pointmlp_bwd_dx_kernel<1>), dim3(grid), dim3(256), sh, st, a, arg, w, Co, seg, (int)S, chunks, dx, lddx);
    hipLaunchKernelGGL((pointmlp_bwd_dw_kernel<1>), g2, dim3(256), 0, st, a, arg, x, ldx, Co, seg, (int)S, nsc, ws);
  }
  SUG_LAUNCH_CHECK("sug_pointmlp_max_bwd_sparse");
  const int64_t n = (int64_t)Co * K;
  hipLaunchKernelGGL(pointmlp_bwd_dw_fold_kernel, dim3(sug_divup(n, 16)), dim3(256), 0, st, ws, nsc, n, dw);
  SUG_LAUNCH_CHECK("sug_pointmlp_max_bwd_sparse(fold)");
  return SUG_OK;
}
