// Point Transformer vector attention, forward, as ONE kernel on the fp16 matrix cores (round 6; BASELINE config 5:
// "fp16 with MFMA attention path").  Replaces, in the 16-bit mode, the chain
//     pos1 -> GEMM(fc_delta[2]) -> qk -> GEMM(fc_gamma[0]) + ReLU -> GEMM(fc_gamma[2]) -> softmax over k + weighted sum
// of TransformerBlock.forward (model/Ptran_transformer.py:39-44), i.e. sug_ptran_pos1_fwd, three library GEMMs on the
// k-expanded rows, sug_ptran_qk_fwd and sug_ptran_attn_fwd with their five [B n k, 512] intermediates round-tripping HBM
// (11 GB of traffic per block at config 5's 1 M rows).
//
// Workgroup = 8 points x 16 neighbours = 128 k-expanded rows, 512 threads = 8 waves, one workgroup per CU:
//   * the 128 x 512 fp16 activations of the current layer live in LDS (130 KB, row stride 520 halves: conflict-free b128
//     reads of the A operand);
//   * three chained 512 x 512 linears on v_mfma_f32_32x32x16_f16: wave w owns output columns [64 w, 64 w + 64) of every
//     layer (4 row tiles x 2 column tiles = 8 accumulators of 16 registers), the A operand comes from LDS, the WEIGHTS
//     stream from L2 straight into the B operand registers (double-buffered 16-byte pieces; staging them through LDS does
//     not fit beside the activations -- tools/ubench/ptran_chain.hip measured this chain at 856 TFLOP/s);
//   * between the layers the accumulators (+ bias, ReLU) go back to LDS as fp16 -- the same rounding points as the library
//     GEMMs' fp16 outputs -- and a "row pass" (wave = one point, lane = 8 channels, as in ptran.hip) does the elementwise
//     work on whole rows: pos1 in front of the first layer, U = (q - K_nbr) + delta behind it, the softmax over the 16
//     neighbours and the weighted sum of V_nbr + delta behind the last.
// What reaches HBM: delta (it is needed twice: in U and in the weighted sum; the second read is an L2 hit of the wave's
// own store) and mixed / max / sum per point; with SAVE (a backward will follow) also T0, U, T1 and the logits, each
// written ONCE with 16-byte stores and never read by the forward.  The backward (ops._PTranAttention) is unchanged.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int FD = 512;            // d_model
constexpr int FK = 16;             // neighbours per point
constexpr int FPTS = 8;            // points per workgroup
constexpr int FROWS = FPTS * FK;   // 128 rows
constexpr int FLDX = FD + 8;       // halves per LDS row
constexpr int FNT = 512;
constexpr int RB = 8;               // rows per batch of a row pass
constexpr float F_LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ float fbcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ void h8_to_f(const h8& x, float (&v)[8]) {
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = (float)x[u];
}
__device__ __forceinline__ h8 f_to_h8(const float (&v)[8]) {
  h8 x;
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = (_Float16)v[u];          // round to nearest even, as __floats2half2_rn
  return x;
}
__device__ __forceinline__ void ldf8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void stf8(float* __restrict__ p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <bool SAVE>
__global__ __launch_bounds__(FNT) void ptran_fused_fwd_kernel(
    const float* __restrict__ xyz, const int32_t* __restrict__ nbr, const float* __restrict__ q,
    const float* __restrict__ kf, const float* __restrict__ vf, const float* __restrict__ W1,
    const float* __restrict__ b1, const _Float16* __restrict__ W2, const _Float16* __restrict__ b2,
    const _Float16* __restrict__ Wg1, const _Float16* __restrict__ bg1, const _Float16* __restrict__ Wg2,
    const _Float16* __restrict__ bg2, int64_t P, int n, float scale, _Float16* __restrict__ T0,
    _Float16* __restrict__ delta, _Float16* __restrict__ U, _Float16* __restrict__ T1, _Float16* __restrict__ Lg,
    float* __restrict__ mixed, float* __restrict__ mx, float* __restrict__ sm) {
  extern __shared__ __attribute__((aligned(16))) _Float16 sx[];       // [FROWS][FLDX]
  __shared__ int s_m[FROWS];                                          // absolute point row (b n + m) of every row's neighbour
  const int t = threadIdx.x, lane = t & 63, j = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t nblk = P / FPTS;

  // one linear layer of the chain: acc = sx . Wl^T for this wave's 64 columns (A from LDS, B streamed from L2)
  auto gemm = [&](const _Float16* __restrict__ Wl, f16v (&acc)[4][2]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
    const _Float16* wp0 = Wl + (size_t)(w * 64 + j) * FD + 8 * h;
    const _Float16* wp1 = wp0 + (size_t)32 * FD;
    h8 bA[2], bB[2];
    bA[0] = *reinterpret_cast<const h8*>(wp0);
    bB[0] = *reinterpret_cast<const h8*>(wp1);
#pragma unroll 2
    for (int s = 0; s < FD / 16; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < FD / 16) {
        bA[nxt] = *reinterpret_cast<const h8*>(wp0 + 16 * (s + 1));
        bB[nxt] = *reinterpret_cast<const h8*>(wp1 + 16 * (s + 1));
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const h8 a = *reinterpret_cast<const h8*>(sx + (rt * 32 + j) * FLDX + 16 * s + 8 * h);
        acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bA[cur], acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bB[cur], acc[rt][1], 0, 0, 0);
      }
    }
  };
  // accumulators (+ bias, optional ReLU) -> fp16 -> LDS, in place of the layer's input (callers put the barriers around it)
  auto to_lds = [&](const f16v (&acc)[4][2], const _Float16* __restrict__ bias, bool relu) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = w * 64 + ct * 32 + j;
      const float bj = (float)bias[col];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rt * 32 + 8 * (r >> 2) + (r & 3) + 4 * h;
          float v = acc[rt][ct][r] + bj;
          if (relu) v = v > 0.f ? v : 0.f;
          sx[row * FLDX + col] = (_Float16)v;
        }
    }
  };

  // W1 | b1 as [512][4] floats in LDS (behind the activation tile), read once per octet by every lane
  float* s_w1 = reinterpret_cast<float*>(sx + FROWS * FLDX);
  for (int c = t; c < FD; c += FNT) {
    *reinterpret_cast<float4*>(s_w1 + c * 4) = make_float4(W1[c * 3 + 0], W1[c * 3 + 1], W1[c * 3 + 2], b1[c]);
  }
  // neighbour offsets of the first octet
  float ndx = 0.f, ndy = 0.f, ndz = 0.f;
  int nm = 0;
  if (lane < FK && (int64_t)blockIdx.x < nblk) {
    const int64_t p0 = (int64_t)blockIdx.x * FPTS + w, pb0 = (p0 / n) * n;
    const int m = nbr[p0 * FK + lane];
    const float* xi = xyz + p0 * 3;
    const float* xj = xyz + (pb0 + m) * 3;
    ndx = xi[0] - xj[0]; ndy = xi[1] - xj[1]; ndz = xi[2] - xj[2];
    nm = (int)(pb0 + m);
  }
  __syncthreads();
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t p = blk * FPTS + w;                  // this wave's point in every row pass
    const int64_t r0 = p * FK;                         // its first k-expanded row
    _Float16* srow = sx + (w * FK) * FLDX + lane * 8;  // the lane's 8 channels of the wave's first row in LDS
    // ---- row pass 0 (pos1): T0 = relu(W1 . (xyz_i - xyz_nbr) + b1) -> LDS (+ T0).  The neighbour offsets of this octet were
    // fetched during the previous octet (ndx / ndy / ndz / nm: the dependent nbr -> xyz loads are the only global latency of
    // this pass and nothing else of the workgroup can run beside them); W1 / b1 come from their LDS copy.
    {
      float wx[8], wy[8], wz[8], bb[8];
      {
        const float4* wl = reinterpret_cast<const float4*>(s_w1 + lane * 32);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 v4 = wl[u];
          wx[u] = v4.x; wy[u] = v4.y; wz[u] = v4.z; bb[u] = v4.w;
        }
      }
      const float dx = ndx, dy = ndy, dz = ndz;
      if (lane < FK) s_m[w * FK + lane] = nm;
      {                                                  // prefetch for the next octet of this workgroup
        const int64_t pn = (blk + gridDim.x) * FPTS + w;
        if (lane < FK && blk + gridDim.x < nblk) {
          const int64_t pbn = (pn / n) * n;
          const int m = nbr[pn * FK + lane];
          const float* xi = xyz + pn * 3;
          const float* xj = xyz + (pbn + m) * 3;
          ndx = xi[0] - xj[0]; ndy = xi[1] - xj[1]; ndz = xi[2] - xj[2];
          nm = (int)(pbn + m);
        }
      }
#pragma unroll
      for (int jj = 0; jj < FK; ++jj) {
        const float ex = fbcast(dx, jj), ey = fbcast(dy, jj), ez = fbcast(dz, jj);
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float tt = fmaf(wz[u], ez, fmaf(wy[u], ey, wx[u] * ex)) + bb[u];
          v[u] = tt > 0.f ? tt : 0.f;
        }
        const h8 hv = f_to_h8(v);
        *reinterpret_cast<h8*>(srow + jj * FLDX) = hv;
        if (SAVE) *reinterpret_cast<h8*>(T0 + (r0 + jj) * FD + lane * 8) = hv;
      }
    }
    __syncthreads();
    // ---- layer 1: delta = T0 . W2^T + b2
    {
      f16v acc[4][2];
      gemm(W2, acc);
      __syncthreads();                                 // every wave has read T0
      to_lds(acc, b2, false);
    }
    __syncthreads();
    // ---- row pass 1 (qk): delta -> HBM; U = (q_i - K_nbr) + delta -> LDS (+ U)
    {
      float qv[8];
      ldf8(q + p * FD + lane * 8, qv);
#pragma unroll 1
      for (int j0 = 0; j0 < FK; j0 += RB) {            // RB rows' gathers in flight per trip (the pass is latency-bound)
        float kv[RB][8];
        h8 dr[RB];
#pragma unroll
        for (int tt = 0; tt < RB; ++tt) {
          ldf8(kf + (int64_t)s_m[w * FK + j0 + tt] * FD + lane * 8, kv[tt]);
          dr[tt] = *reinterpret_cast<const h8*>(srow + (j0 + tt) * FLDX);
        }
#pragma unroll
        for (int tt = 0; tt < RB; ++tt) {
          *reinterpret_cast<h8*>(delta + (r0 + j0 + tt) * FD + lane * 8) = dr[tt];
          float dv[8], o[8];
          h8_to_f(dr[tt], dv);
#pragma unroll
          for (int u = 0; u < 8; ++u) o[u] = (qv[u] - kv[tt][u]) + dv[u];
          const h8 hv = f_to_h8(o);
          *reinterpret_cast<h8*>(srow + (j0 + tt) * FLDX) = hv;
          if (SAVE) *reinterpret_cast<h8*>(U + (r0 + j0 + tt) * FD + lane * 8) = hv;
        }
      }
    }
    __syncthreads();
    // ---- layer 2: T1 = relu(U . Wg1^T + bg1)
    {
      f16v acc[4][2];
      gemm(Wg1, acc);
      __syncthreads();
      to_lds(acc, bg1, true);
    }
    __syncthreads();
    if (SAVE) {
#pragma unroll
      for (int jj = 0; jj < FK; ++jj)
        *reinterpret_cast<h8*>(T1 + (r0 + jj) * FD + lane * 8) = *reinterpret_cast<const h8*>(srow + jj * FLDX);
    }
    // ---- layer 3: L = T1 . Wg2^T + bg2
    {
      f16v acc[4][2];
      gemm(Wg2, acc);
      __syncthreads();
      to_lds(acc, bg2, false);
    }
    __syncthreads();
    // ---- row pass 3 (attn): softmax over the 16 neighbours of L * scale (per channel), applied to V_nbr + delta
    {
      float zmax[8], zsum[8], av[8], zl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { zmax[u] = -INFINITY; zsum[u] = 0.f; av[u] = 0.f; }
#pragma unroll 4
      for (int jj = 0; jj < FK; ++jj) {
        const h8 z = *reinterpret_cast<const h8*>(srow + jj * FLDX);
        if (SAVE) *reinterpret_cast<h8*>(Lg + (r0 + jj) * FD + lane * 8) = z;
        float zv[8];
        h8_to_f(z, zv);
#pragma unroll
        for (int u = 0; u < 8; ++u) zmax[u] = fmaxf(zmax[u], zv[u] * scale);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) zl[u] = -zmax[u] * F_LOG2E;
#pragma unroll 1
      for (int j0 = 0; j0 < FK; j0 += RB) {
        float vv[RB][8];
        h8 dr[RB], zr[RB];
#pragma unroll
        for (int tt = 0; tt < RB; ++tt) {
          ldf8(vf + (int64_t)s_m[w * FK + j0 + tt] * FD + lane * 8, vv[tt]);
          dr[tt] = *reinterpret_cast<const h8*>(delta + (r0 + j0 + tt) * FD + lane * 8);   // this lane's own store of pass 1
          zr[tt] = *reinterpret_cast<const h8*>(srow + (j0 + tt) * FLDX);                  // (the logits a second time, from LDS)
        }
#pragma unroll
        for (int tt = 0; tt < RB; ++tt) {
          float zv[8], dv[8];
          h8_to_f(zr[tt], zv);
          h8_to_f(dr[tt], dv);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float e = __builtin_amdgcn_exp2f(fmaf(zv[u], scale * F_LOG2E, zl[u]));
            zsum[u] += e;
            av[u] = fmaf(e, vv[tt][u] + dv[u], av[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) av[u] /= zsum[u];
      stf8(mixed + p * FD + lane * 8, av);
      stf8(mx + p * FD + lane * 8, zmax);
      stf8(sm + p * FD + lane * 8, zsum);
    }
    __syncthreads();                                   // the next octet's pos1 overwrites the tile
  }
}

}  // namespace

// 1 if sug_ptran_fused_fwd takes this shape: d_model 512, exactly 16 neighbours, whole octets of points
extern "C" int sug_ptran_fused_supported(int B, int n, int k, int d) {
  return (d == FD && k == FK && B > 0 && n > 0 && ((int64_t)B * n) % FPTS == 0) ? 1 : 0;
}

extern "C" int sug_ptran_fused_fwd(const float* xyz, const int32_t* nbr, const float* q, const float* kf, const float* vf,
                                   const float* w1, const float* b1, const void* w2, const void* b2, const void* wg1,
                                   const void* bg1, const void* wg2, const void* bg2, int B, int n, int k, int d,
                                   float scale, int save, void* T0, void* delta, void* U, void* T1, void* Lg, float* mixed,
                                   float* mx, float* sm, void* stream) {
  SUG_REQUIRE(xyz && nbr && q && kf && vf && w1 && b1 && w2 && b2 && wg1 && bg1 && wg2 && bg2 && delta && mixed && mx && sm,
              "sug_ptran_fused_fwd: null pointer");
  SUG_REQUIRE(sug_ptran_fused_supported(B, n, k, d), "sug_ptran_fused_fwd: needs d_model 512, k = 16 and B*n a multiple of 8");
  SUG_REQUIRE(!save || (T0 && U && T1 && Lg), "sug_ptran_fused_fwd: save = 1 needs the T0 / U / T1 / logits buffers");
  SUG_REQUIRE((int64_t)B * n < (1ll << 31), "sug_ptran_fused_fwd: too many points");
  const int64_t P = (int64_t)B * n;
  const size_t sh = (size_t)FROWS * FLDX * sizeof(_Float16) + (size_t)FD * 4 * sizeof(float);     // activation tile + W1 | b1
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = P / FPTS;
  const int ncu = sug_cu_count();
  const int grid = (int)(nblk < 2 * (int64_t)ncu ? nblk : 2 * (int64_t)ncu);     // one 130 KB workgroup per CU, grid-stride over the octets
  const _Float16 *W2 = (const _Float16*)w2, *B2 = (const _Float16*)b2, *WG1 = (const _Float16*)wg1, *BG1 = (const _Float16*)bg1,
                 *WG2 = (const _Float16*)wg2, *BG2 = (const _Float16*)bg2;
  if (save) {
    static SugLdsOptIn note;
    if (int rc = sug_allow_dynamic_lds(note, &ptran_fused_fwd_kernel<true>, (int)sh, "sug_ptran_fused_fwd")) return rc;
    hipLaunchKernelGGL(ptran_fused_fwd_kernel<true>, dim3(grid), dim3(FNT), sh, st, xyz, nbr, q, kf, vf, w1, b1, W2, B2, WG1, BG1, WG2,
                       BG2, P, n, scale, (_Float16*)T0, (_Float16*)delta, (_Float16*)U, (_Float16*)T1, (_Float16*)Lg, mixed, mx, sm);
  } else {
    static SugLdsOptIn note;
    if (int rc = sug_allow_dynamic_lds(note, &ptran_fused_fwd_kernel<false>, (int)sh, "sug_ptran_fused_fwd")) return rc;
    hipLaunchKernelGGL(ptran_fused_fwd_kernel<false>, dim3(grid), dim3(FNT), sh, st, xyz, nbr, q, kf, vf, w1, b1, W2, B2, WG1, BG1, WG2,
                       BG2, P, n, scale, (_Float16*)nullptr, (_Float16*)delta, (_Float16*)nullptr, (_Float16*)nullptr,
                       (_Float16*)nullptr, mixed, mx, sm);
  }
  SUG_LAUNCH_CHECK("sug_ptran_fused_fwd");
  return SUG_OK;
}
