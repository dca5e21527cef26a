// Classifier heads (Pointnet_c, model/Model.py:412-449; fc_layer, model/model_utils.py:35-57) for a handful of rows:
//     x [M, K1] -> Linear -> LayerNorm -> act -> Dropout -> Linear -> LayerNorm -> act (= mid feature) -> Dropout -> Linear
// with M = the clouds of a step (<= 128) and both heads of Net_MDA (c1, c2: same input, own weights) in ONE launch per
// layer.  The library path is ~47 launches of 5-30 us per step for the two heads (three GEMMs with M = 64, bias sums,
// LayerNorm / activation / dropout kernels and their backwards, gradient adds) -- launch latency, not work: the whole
// arithmetic is 0.17 GFLOP and 5 MB of weights.  Here:
//   forward   3 launches  head_fwd_kernel:  z = pro(in) . W^T + b; pro = Dropout(act(LayerNorm(.))) of the PREVIOUS layer,
//             applied to the A operand as it is loaded (every workgroup forms the row statistics of its M input rows itself);
//   backward  3 launches  head_bwd_kernel:  dz = LayerNorm/act/Dropout backward of THIS layer's output gradient (again per
//             workgroup, the rows are short), then dW = dz^T . pro(in), db, dgamma, dbeta and the input gradient dz . W.
// fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32); every sum has a fixed order (no atomics).
//
// Operand mapping of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// register r of the result holds D[i = 8 (r >> 2) + (r & 3) + 4 (l >> 5)][j = l & 31].  Which two k's a step multiplies
// is free as long as A and B agree: lane half h takes a CONTIGUOUS run of k, so operands are plain float4 loads.
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HMAX = 2;          // heads per launch
constexpr int NT = 512;          // threads per workgroup (8 waves)
constexpr int NW = NT / 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct FwdHead {
  const float* in;      // [M, K]  x, or the previous layer's pre-LayerNorm output when PRO
  const float* W;       // [No, K]
  const float* bias;    // [No] or null
  float* z;             // [M, No]
  const float* gamma;   // [K]  LayerNorm of the previous layer (PRO)
  const float* beta;
  const float* u;       // [M, K] uniform randoms of the dropout behind that LayerNorm, or null (no dropout)
  float* stats;         // [M, 2] mean | rstd of the rows of `in` (PRO; written by workgroup x == 0), or null
  float* act;           // [M, K] act(LayerNorm(in)) BEFORE the dropout (the mid feature), or null (workgroup x == 0)
};
struct FwdArgs {
  FwdHead h[HMAX];
  int64_t ldin;
  int M, K, No;
  float slope, eps, p, keep;      // dropout: a value survives when u >= p and is scaled by keep = 1 / (1 - p)
};

// a = Dropout(act(LayerNorm(v))) for one element; `pre` receives the value before the dropout
__device__ __forceinline__ float pro_elem(float v, float mean, float rstd, float g, float b, float slope, float& pre) {
  const float y = fmaf((v - mean) * rstd, g, b);
  pre = y > 0.f ? y : slope * y;
  return pre;
}

// KT: K at compile time (256, 512, 1024: the loop over the k chunks is straight-line code and the compiler issues the
// operand loads of many chunks before the first MFMA -- with a runtime loop every chunk paid its own L2 latency, 20 us for a
// 7 us kernel) or 0 (any K % 64 == 0).
template <bool PRO, int RB, int KT>
__global__ __launch_bounds__(NT) void head_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_stats = smem;                       // [RB * 32][2]
  float* s_red = smem + RB * 64;               // [NW][RB * 32][32]
  const FwdHead H = a.h[blockIdx.y];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  const int M = a.M, K = KT ? KT : a.K, No = a.No;
  const int ldin = (int)a.ldin;
  const int n0 = blockIdx.x * 32;
  if constexpr (PRO) {
    // row statistics of `in` (two passes over registers, as torch's LayerNorm: mean, then the mean of squared deviations);
    // a wave takes 8 rows at a time with all their loads in flight together (K <= 1024: 4 float4 per lane and row)
    constexpr int RPW = 8;
    for (int g0 = 0; g0 < RB * 32; g0 += NW * RPW) {
      float4 v[RPW][4];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const float* row = H.in + min(g0 + w * RPW + r, M - 1) * ldin + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = (c * 256 < K) ? ld4(row + c * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) sum += (v[r][c].x + v[r][c].y) + (v[r][c].z + v[r][c].w);
        const float mean = wave_sum_f(sum) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c * 256 < K) {
            const float dx = v[r][c].x - mean, dy = v[r][c].y - mean, dz = v[r][c].z - mean, dw = v[r][c].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
          }
        }
        const float rstd = 1.0f / sqrtf(wave_sum_f(q) / (float)K + a.eps);
        const int i = g0 + w * RPW + r;
        if (lane == 0) {
          s_stats[2 * i] = mean; s_stats[2 * i + 1] = rstd;
          if (blockIdx.x == 0 && i < M && H.stats) { H.stats[2 * i] = mean; H.stats[2 * i + 1] = rstd; }
        }
      }
    }
    __syncthreads();
  }
  // wave w multiplies the k range [w K/8, (w+1) K/8): lane half h its contiguous half of it, 4 k per chunk
  const int KH = K / (2 * NW);                 // k per lane half
  const int kb = w * (K / NW) + h * KH;
  const float* wrow = H.W + min(n0 + j, No - 1) * K + kb;
  const float* arow[RB];
  float mean[RB], rstd[RB];
  int irow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    irow[rb] = min(rb * 32 + j, M - 1);
    arow[rb] = H.in + irow[rb] * ldin + kb;
    if constexpr (PRO) { mean[rb] = s_stats[2 * (rb * 32 + j)]; rstd[rb] = s_stats[2 * (rb * 32 + j) + 1]; }
  }
  const bool write_act = PRO && blockIdx.x == 0 && H.act != nullptr;
  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  // The operand loads of a BATCH of chunks are issued together, then the batch is multiplied: one L2 round trip per
  // batch (left to itself the compiler keeps one or two chunks in flight and every chunk pays the latency).
  constexpr int CB = PRO ? 4 : (RB <= 2 ? 8 : 4);       // chunks per batch: 28 / 12 / 20 registers per chunk
  auto batch = [&](int c0, int nch) {                   // chunks c0, c0 + 4, ...: nch of them (<= CB)
    float4 b4[CB], a4[CB][RB], g4[PRO ? CB : 1], be4[PRO ? CB : 1], u4[PRO ? CB : 1][RB];
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      if (q < nch) {
        const int c = c0 + 4 * q;
        b4[q] = ld4(wrow + c);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a4[q][rb] = ld4(arow[rb] + c);
        if constexpr (PRO) {
          g4[q] = ld4(H.gamma + kb + c); be4[q] = ld4(H.beta + kb + c);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
            u4[q][rb] = H.u ? ld4(H.u + irow[rb] * K + kb + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);        // (keeps the loads above the arithmetic: the scheduler sinks them otherwise)
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      if (q < nch) {
        const int c = c0 + 4 * q;
        if constexpr (PRO) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            float4 pre;
            float4 v = a4[q][rb];
            v.x = pro_elem(v.x, mean[rb], rstd[rb], g4[q].x, be4[q].x, a.slope, pre.x);
            v.y = pro_elem(v.y, mean[rb], rstd[rb], g4[q].y, be4[q].y, a.slope, pre.y);
            v.z = pro_elem(v.z, mean[rb], rstd[rb], g4[q].z, be4[q].z, a.slope, pre.z);
            v.w = pro_elem(v.w, mean[rb], rstd[rb], g4[q].w, be4[q].w, a.slope, pre.w);
            if (write_act && rb * 32 + j < M) *reinterpret_cast<float4*>(H.act + irow[rb] * K + kb + c) = pre;
            if (H.u) {
              const float4 uu = u4[q][rb];
              v.x = uu.x >= a.p ? v.x * a.keep : 0.f; v.y = uu.y >= a.p ? v.y * a.keep : 0.f;
              v.z = uu.z >= a.p ? v.z * a.keep : 0.f; v.w = uu.w >= a.p ? v.w * a.keep : 0.f;
            }
            a4[q][rb] = v;
          }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][rb].x, b4[q].x, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][rb].y, b4[q].y, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][rb].z, b4[q].z, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][rb].w, b4[q].w, acc[rb], 0, 0, 0);
        }
      }
    }
  };
  if constexpr (KT != 0) {
    constexpr int NCH = KT / (8 * NW);                   // chunks per lane half
#pragma unroll
    for (int q0 = 0; q0 < NCH; q0 += CB) {
      batch(4 * q0, NCH - q0 < CB ? NCH - q0 : CB);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const int nchunk = KH >> 2;
    for (int q0 = 0; q0 < nchunk; q0 += CB) batch(4 * q0, min(CB, nchunk - q0));
  }
  // the 8 partial tiles -> LDS -> ordered sum + bias
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = rb * 32 + 8 * (r >> 2) + (r & 3) + 4 * h;
      s_red[(w * RB * 32 + i) * 32 + j] = acc[rb][r];
    }
  __syncthreads();
  for (int e = t; e < RB * 32 * 32; e += NT) {
    const int i = e >> 5, jj = e & 31;
    float s = s_red[e];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) s += s_red[ww * RB * 32 * 32 + e];
    if (i < M && n0 + jj < No) H.z[i * No + n0 + jj] = s + (H.bias ? H.bias[n0 + jj] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------- backward
struct BwdHead {
  const float* gup;      // [M, No] (row stride ldg): gradient of this layer's output AFTER LayerNorm / act / dropout (EPI) or of z itself
  const float* z;        // EPI: this layer's pre-LayerNorm output [M, No], its row statistics [M, 2], LayerNorm weights [No],
  const float* stats;    //      dropout randoms [M, No] or null, extra gradient of the activation before the dropout [M, No] or null
  const float* gamma;
  const float* beta;
  const float* u;
  const float* gextra;
  const float* in;       // [M, K] (row stride ldin): x, or (PRO) the previous layer's z with its statistics / LayerNorm / dropout
  const float* stats_in;
  const float* gamma_in;
  const float* beta_in;
  const float* u_in;
  const float* W;        // [No, K]
  float* dW;             // [No, K]
  float* db;             // [No] or null
  float* dgamma;         // [No] (EPI)
  float* dbeta;
};
struct BwdArgs {
  BwdHead h[HMAX];
  float* da;             // sum_da: one [M, K] input gradient summed over the heads; else per head in da2
  float* da2[HMAX];
  int ldg, ldin, ldda;   // (row strides: 32-bit index arithmetic throughout -- every operand here is a few MB at most)
  int M, K, No, NoP, heads, sum_da;
  float slope, eps, p, keep, p_in, keep_in;     // p: dropout behind this layer's LayerNorm; p_in: behind the input's
};

// dz of one row block (32 rows) into s_dz[32][NOP + 1] (columns >= No: zero).  16 threads per row.
template <bool EPI, int NOP>
__device__ __forceinline__ void dz_block(const BwdHead& H, const BwdArgs& a, int r0, float* __restrict__ s_dz) {
  const int t = threadIdx.x, i = t >> 4, q = t & 15;
  const int M = a.M, No = a.No;
  constexpr int RS = NOP + 1;
  const int row = r0 + i;
  const bool ok = row < M;
  const int rc = ok ? row : M - 1;
  if constexpr (!EPI) {
#pragma unroll
    for (int c0 = 0; c0 < NOP; c0 += 16) {
      const int c = c0 + q;
      s_dz[i * RS + c] = (ok && c < No) ? H.gup[rc * a.ldg + min(c, No - 1)] : 0.f;
    }
  } else {
    // No = NOP is a multiple of 64: float4 pieces, column (q + 16 u) * 4.  Pass 1 (4 pieces' loads in flight together)
    // leaves d = dL/d(zhat) in s_dz and the row sums in registers, pass 2 re-reads z (one load per piece) and finishes
    // dz = rstd (d - mean(d) - zhat mean(d zhat)).  Straight-line code: a loop that waited for every piece's loads in
    // turn cost 16 L2 round trips per row block.
    const float mean = H.stats[2 * rc], rstd = H.stats[2 * rc + 1];
    constexpr int NU = NOP >> 6;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = (q + 16 * u) * 4;
      const float4 z4 = ld4(H.z + rc * No + c), g4 = ld4(H.gamma + c), b4 = ld4(H.beta + c);
      float4 gu = ld4(H.gup + rc * a.ldg + c);
      if (H.u) {
        const float4 u4 = ld4(H.u + rc * No + c);
        gu.x = u4.x >= a.p ? gu.x * a.keep : 0.f; gu.y = u4.y >= a.p ? gu.y * a.keep : 0.f;
        gu.z = u4.z >= a.p ? gu.z * a.keep : 0.f; gu.w = u4.w >= a.p ? gu.w * a.keep : 0.f;
      }
      if (H.gextra) {
        const float4 e4 = ld4(H.gextra + rc * No + c);
        gu.x += e4.x; gu.y += e4.y; gu.z += e4.z; gu.w += e4.w;
      }
      float4 zz, dd;
      zz.x = (z4.x - mean) * rstd; zz.y = (z4.y - mean) * rstd; zz.z = (z4.z - mean) * rstd; zz.w = (z4.w - mean) * rstd;
      dd.x = (fmaf(zz.x, g4.x, b4.x) > 0.f ? gu.x : a.slope * gu.x) * g4.x;
      dd.y = (fmaf(zz.y, g4.y, b4.y) > 0.f ? gu.y : a.slope * gu.y) * g4.y;
      dd.z = (fmaf(zz.z, g4.z, b4.z) > 0.f ? gu.z : a.slope * gu.z) * g4.z;
      dd.w = (fmaf(zz.w, g4.w, b4.w) > 0.f ? gu.w : a.slope * gu.w) * g4.w;
      s1 += (dd.x + dd.y) + (dd.z + dd.w);
      s2 += (dd.x * zz.x + dd.y * zz.y) + (dd.z * zz.z + dd.w * zz.w);
      float* d = s_dz + i * RS + c;
      d[0] = dd.x; d[1] = dd.y; d[2] = dd.z; d[3] = dd.w;
      if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);         // four pieces' loads in flight, not all eight (registers)
    }
    // sums over the 16 threads of the row (xor tree: fixed order)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 / (float)No, m2 = s2 / (float)No;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = (q + 16 * u) * 4;
      const float4 z4 = ld4(H.z + rc * No + c);
      float* d = s_dz + i * RS + c;                       // (this thread's own pieces: no barrier needed)
      d[0] = ok ? rstd * (d[0] - m1 - (z4.x - mean) * rstd * m2) : 0.f;
      d[1] = ok ? rstd * (d[1] - m1 - (z4.y - mean) * rstd * m2) : 0.f;
      d[2] = ok ? rstd * (d[2] - m1 - (z4.z - mean) * rstd * m2) : 0.f;
      d[3] = ok ? rstd * (d[3] - m1 - (z4.w - mean) * rstd * m2) : 0.f;
    }
  }
}

// one element of the layer input a = pro(in)
template <bool PRO>
__device__ __forceinline__ float in_elem(const BwdHead& H, const BwdArgs& a, int row, int k) {
  const float v = H.in[row * a.ldin + k];
  if constexpr (!PRO) return v;
  const float y = fmaf((v - H.stats_in[2 * row]) * H.stats_in[2 * row + 1], H.gamma_in[k], H.beta_in[k]);
  float r = y > 0.f ? y : a.slope * y;
  if (H.u_in) r = H.u_in[row * a.K + k] >= a.p_in ? r * a.keep_in : 0.f;
  return r;
}

// Workgroup = 32 input columns [k0, k0+32) (x heads unless sum_da).  NOP: output width padded to 32 (32, 256 or 512:
// compile-time, so that the MFMA loops are straight-line code with static register indices).
template <bool EPI, bool PRO, int NOP>
__global__ __launch_bounds__(NT) void head_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NoP = NOP, RS = NoP + 1;
  constexpr int MAXT = (NOP / 32 + NW - 1) / NW;       // dW tiles (32 output rows each) per wave
  const int M = a.M, K = a.K, No = a.No;
  float* s_dz = smem;                          // [32][NoP + 1]
  float* s_a = s_dz + 32 * RS;                 // [32][33]
  float* s_red = s_a + 32 * 33;                // [NW][32][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * 32;
  const int hd0 = a.sum_da ? 0 : blockIdx.y, hd1 = a.sum_da ? a.heads : blockIdx.y + 1;
  constexpr int ntile = NoP >> 5;              // dW tiles of 32 output rows; wave w owns tiles w, w + 8, ...
  constexpr int NWK = NoP / NW;                // output rows per wave in the dz . W product (K dimension split over the waves)
  constexpr int nh = NWK >> 1;                 // ... per lane half
  const int nblk = (M + 31) >> 5;
  const int cpw = (No + (int)gridDim.x - 1) / (int)gridDim.x;      // parameter-gradient columns per workgroup
  float dsum[2 * 4];                           // sum_da: this thread's 2 elements of every row block (<= 4), over the heads
#pragma unroll
  for (int e = 0; e < 8; ++e) dsum[e] = 0.f;

  for (int hd = hd0; hd < hd1; ++hd) {
    const BwdHead H = a.h[hd];
    f32x16 accw[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[q][r] = 0.f;
    float gsum = 0.f, bsum = 0.f, dbs = 0.f;   // threads 0..15: dgamma / dbeta / db of column blockIdx.x * cpw + t

    for (int blk = 0; blk < nblk; ++blk) {
      const int r0 = blk * 32;
      __syncthreads();                          // the previous block's readers of s_dz / s_a / s_red are done
      dz_block<EPI, NOP>(H, a, r0, s_dz);
      // W[n][k0 + j] for this wave's n range (lane half h: the contiguous half) -- issued here, used by the last phase
      // of the block: the L2 round trip hides under the phases between (re-read per row block: registers, not bytes,
      // are what this kernel is short of)
      float wreg[nh];
#pragma unroll
      for (int s = 0; s < nh; ++s) {
        const int n = w * NWK + h * nh + s;
        wreg[s] = H.W[min(n, No - 1) * K + k0 + j];      // (rows >= No meet dz columns that are zero)
      }
      {
        // a[r0 .. r0+32)[k0 .. k0+32): two elements per thread
        const int i = t >> 4, c = (t & 15) * 2;
        const int row = r0 + i;
        const bool ok = row < M;
        const int rc = ok ? row : M - 1;
        s_a[i * 33 + c] = ok ? in_elem<PRO>(H, a, rc, k0 + c) : 0.f;
        s_a[i * 33 + c + 1] = ok ? in_elem<PRO>(H, a, rc, k0 + c + 1) : 0.f;
      }
      __syncthreads();
      // ---- parameter gradients of the LayerNorm and the bias: workgroup x owns the columns [x cpw, (x+1) cpw), cpw <= 16
      // (host-checked).  Thread (row i, column cc) forms its one term, 16 threads then add the 32 rows in ascending order.
      {
        const int i = t >> 4, cc = t & 15;
        const int c = blockIdx.x * cpw + cc, row = r0 + i;
        const bool on = cc < cpw && c < No && row < M;
        float pg = 0.f, pb = 0.f, pd = 0.f;
        if (on) {
          pd = s_dz[i * RS + c];
          if constexpr (EPI) {
            const float zz = (H.z[row * No + c] - H.stats[2 * row]) * H.stats[2 * row + 1];
            float gu = H.gup[row * a.ldg + c];
            if (H.u) gu = H.u[row * No + c] >= a.p ? gu * a.keep : 0.f;
            if (H.gextra) gu += H.gextra[row * No + c];
            const float g = fmaf(zz, H.gamma[c], H.beta[c]) > 0.f ? gu : a.slope * gu;
            pg = g * zz;
            pb = g;
          }
        }
        s_red[t] = pg; s_red[NT + t] = pb; s_red[2 * NT + t] = pd;
        __syncthreads();
        if (t < 16) {
#pragma unroll 8
          for (int r = 0; r < 32; ++r) { gsum += s_red[r * 16 + t]; bsum += s_red[NT + r * 16 + t]; dbs += s_red[2 * NT + r * 16 + t]; }
        }
        __syncthreads();
      }
      // ---- dW[n][k0 + j] += sum_i dz[i][n] a[i][k0 + j]: A = dz^T (lane = n), B = a (lane = column), k = the block's rows
#pragma unroll
      for (int q = 0; q < MAXT; ++q) {
        const int tile = w + q * NW;
        if (tile < ntile) {
          const float* ap = s_dz + tile * 32 + j;
#pragma unroll
          for (int s = 0; s < 16; ++s)
            accw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(2 * s + h) * RS], s_a[(2 * s + h) * 33 + j], accw[q], 0, 0, 0);
        }
      }
      // ---- da[i][k0 + j] = sum_n dz[i][n] W[n][k0 + j]: A = dz (lane = row), B = W (registers), n split over the waves
      if (a.da || a.da2[hd]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = s_dz + j * RS + w * NWK + h * nh;
#pragma unroll
        for (int s = 0; s < nh; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s], wreg[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[(w * 32 + 8 * (r >> 2) + (r & 3) + 4 * h) * 32 + j] = acc[r];
        __syncthreads();
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int e = t + e2 * NT;
          const int i = e >> 5, jj = e & 31;
          float s = s_red[e];
#pragma unroll
          for (int ww = 1; ww < NW; ++ww) s += s_red[ww * 1024 + e];
          if (a.sum_da) {
            // (static register index: blk < 4)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
              if (bb == blk) dsum[2 * bb + e2] += s;
            if (hd == hd1 - 1 && r0 + i < M) {
              float v = 0.f;
#pragma unroll
              for (int bb = 0; bb < 4; ++bb)
                if (bb == blk) v = dsum[2 * bb + e2];
              a.da[(r0 + i) * a.ldda + k0 + jj] = v;
            }
          } else if (r0 + i < M) {
            a.da2[hd][(r0 + i) * a.ldda + k0 + jj] = s;
          }
        }
      }
    }
    // ---- store dW tiles, db / dgamma / dbeta
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const int tile = w + q * NW;
      if (tile < ntile) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = tile * 32 + 8 * (r >> 2) + (r & 3) + 4 * h;
          if (n < No) H.dW[n * K + k0 + j] = accw[q][r];
        }
      }
    }
    if (t < cpw && blockIdx.x * cpw + t < No) {
      const int c = blockIdx.x * cpw + t;
      if (H.db) H.db[c] = dbs;
      if constexpr (EPI) { H.dgamma[c] = gsum; H.dbeta[c] = bsum; }
    }
  }
}

int heads_ok(int heads) { return heads >= 1 && heads <= HMAX; }

}  // namespace

extern "C" int sug_head_linear_supported(int M, int K, int No, int pro, int epi) {
  if (M < 1 || M > 128 || K < 64 || K % 64 || K > 4096 || No < 1) return 0;
  if (!(No <= 32 || No == 256 || No == 512)) return 0;   // (the backward kernel is instantiated for these output widths)
  if (pro && K > 1024) return 0;                    // (row statistics pass: any K % 256 == 0 works; K % 64 for the operand split)
  if (pro && K % 256) return 0;
  if (epi && (No % 64 || No > 512)) return 0;
  return 1;       // (the backward also needs No <= 16 * K / 32: sug_head_linear_bwd checks)
}

extern "C" int sug_head_linear_fwd(int heads, const float* const* in, int64_t ldin, const float* const* W,
                                   const float* const* bias, float* const* z, const float* const* gamma,
                                   const float* const* beta, const float* const* u, float* const* stats,
                                   float* const* act, int M, int K, int No, int pro, float slope, float eps, float p_drop,
                                   void* stream) {
  SUG_REQUIRE(heads_ok(heads), "sug_head_linear_fwd: 1 or 2 heads per launch, got %d", heads);
  SUG_REQUIRE(sug_head_linear_supported(M, K, No, pro, 0), "sug_head_linear_fwd: unsupported shape M=%d K=%d No=%d", M, K, No);
  SUG_REQUIRE(in && W && z && ldin >= K && ldin % 4 == 0, "sug_head_linear_fwd: bad operands");
  SUG_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sug_head_linear_fwd: dropout probability %g", (double)p_drop);
  FwdArgs a;
  for (int hd = 0; hd < heads; ++hd) {
    SUG_REQUIRE(in[hd] && W[hd] && z[hd], "sug_head_linear_fwd: null operand of head %d", hd);
    SUG_REQUIRE(!pro || (gamma && beta && gamma[hd] && beta[hd]), "sug_head_linear_fwd: LayerNorm weights missing");
    SUG_REQUIRE(((uintptr_t)in[hd] % 16) == 0 && ((uintptr_t)W[hd] % 16) == 0, "sug_head_linear_fwd: operands must be 16-byte aligned");
    FwdHead& H = a.h[hd];
    H.in = in[hd]; H.W = W[hd]; H.bias = bias ? bias[hd] : nullptr; H.z = z[hd];
    H.gamma = pro ? gamma[hd] : nullptr; H.beta = pro ? beta[hd] : nullptr;
    H.u = (pro && u) ? u[hd] : nullptr; H.stats = (pro && stats) ? stats[hd] : nullptr; H.act = (pro && act) ? act[hd] : nullptr;
  }
  for (int hd = heads; hd < HMAX; ++hd) a.h[hd] = a.h[0];
  a.ldin = ldin; a.M = M; a.K = K; a.No = No; a.slope = slope; a.eps = eps; a.p = p_drop; a.keep = 1.0f / (1.0f - p_drop);
  hipStream_t st = (hipStream_t)stream;
  const int RB = (M + 31) / 32;
  const dim3 grid((No + 31) / 32, heads), block(NT);
  static SugLdsOptIn note[16];
#define SUG_HEAD_FWD(PRO_, RB_, KT_, SLOT_)                                                                         \
  do {                                                                                                              \
    const size_t sh = (size_t)(RB_ * 64 + NW * RB_ * 32 * 32) * sizeof(float);                                      \
    if (int rc = sug_allow_dynamic_lds(note[SLOT_], &head_fwd_kernel<PRO_, RB_, KT_>, 140 * 1024, "sug_head_linear_fwd")) return rc; \
    hipLaunchKernelGGL((head_fwd_kernel<PRO_, RB_, KT_>), grid, block, sh, st, a);                                  \
  } while (0)
#define SUG_HEAD_FWD_K(PRO_, RB_, SLOT_)                                                                            \
  do {                                                                                                              \
    if (K == 256) SUG_HEAD_FWD(PRO_, RB_, 256, SLOT_);                                                              \
    else if (K == 512) SUG_HEAD_FWD(PRO_, RB_, 512, SLOT_ + 1);                                                     \
    else if (K == 1024) SUG_HEAD_FWD(PRO_, RB_, 1024, SLOT_ + 2);                                                   \
    else SUG_HEAD_FWD(PRO_, RB_, 0, SLOT_ + 3);                                                                     \
  } while (0)
  if (pro) {
    if (RB <= 2) SUG_HEAD_FWD_K(true, 2, 0); else SUG_HEAD_FWD_K(true, 4, 4);
  } else {
    if (RB <= 2) SUG_HEAD_FWD_K(false, 2, 8); else SUG_HEAD_FWD_K(false, 4, 12);
  }
#undef SUG_HEAD_FWD_K
#undef SUG_HEAD_FWD
  SUG_LAUNCH_CHECK("sug_head_linear_fwd");
  return SUG_OK;
}

extern "C" int sug_head_linear_bwd(int heads, int sum_da, const float* const* gup, int64_t ldg, const float* const* z,
                                   const float* const* stats, const float* const* gamma, const float* const* beta,
                                   const float* const* u, const float* const* gextra, const float* const* in, int64_t ldin,
                                   const float* const* stats_in, const float* const* gamma_in, const float* const* beta_in,
                                   const float* const* u_in, const float* const* W, float* const* dW, float* const* db,
                                   float* const* dgamma, float* const* dbeta, float* const* da, int64_t ldda, int M, int K,
                                   int No, int epi, int pro, float slope, float eps, float p_drop, float p_drop_in,
                                   void* stream) {
  SUG_REQUIRE(heads_ok(heads), "sug_head_linear_bwd: 1 or 2 heads per launch, got %d", heads);
  SUG_REQUIRE(sug_head_linear_supported(M, K, No, pro, epi), "sug_head_linear_bwd: unsupported shape M=%d K=%d No=%d", M, K, No);
  SUG_REQUIRE(gup && in && W && dW && ldg >= No && ldin >= K, "sug_head_linear_bwd: bad operands");
  SUG_REQUIRE(No <= 16 * (K / 32), "sug_head_linear_bwd: No=%d needs K >= %d (parameter sums: 16 columns per workgroup)", No, 2 * No);
  SUG_REQUIRE(!epi || (ldg % 4 == 0), "sug_head_linear_bwd: gradient rows must be 16-byte aligned");
  SUG_REQUIRE(p_drop >= 0.f && p_drop < 1.f && p_drop_in >= 0.f && p_drop_in < 1.f,
              "sug_head_linear_bwd: dropout probabilities %g, %g", (double)p_drop, (double)p_drop_in);
  BwdArgs a;
  a.da = nullptr;
  for (int hd = 0; hd < HMAX; ++hd) a.da2[hd] = nullptr;
  for (int hd = 0; hd < heads; ++hd) {
    SUG_REQUIRE(gup[hd] && in[hd] && W[hd] && dW[hd], "sug_head_linear_bwd: null operand of head %d", hd);
    SUG_REQUIRE(!epi || (z && stats && gamma && beta && dgamma && dbeta && z[hd] && stats[hd] && gamma[hd] && beta[hd] &&
                         dgamma[hd] && dbeta[hd]), "sug_head_linear_bwd: LayerNorm operands of the layer missing");
    SUG_REQUIRE(!pro || (stats_in && gamma_in && beta_in && stats_in[hd] && gamma_in[hd] && beta_in[hd]),
                "sug_head_linear_bwd: LayerNorm operands of the input missing");
    BwdHead& H = a.h[hd];
    H.gup = gup[hd];
    H.z = epi ? z[hd] : nullptr; H.stats = epi ? stats[hd] : nullptr; H.gamma = epi ? gamma[hd] : nullptr;
    H.beta = epi ? beta[hd] : nullptr; H.u = (epi && u) ? u[hd] : nullptr; H.gextra = (epi && gextra) ? gextra[hd] : nullptr;
    H.in = in[hd];
    H.stats_in = pro ? stats_in[hd] : nullptr; H.gamma_in = pro ? gamma_in[hd] : nullptr; H.beta_in = pro ? beta_in[hd] : nullptr;
    H.u_in = (pro && u_in) ? u_in[hd] : nullptr;
    H.W = W[hd]; H.dW = dW[hd]; H.db = db ? db[hd] : nullptr;
    H.dgamma = epi ? dgamma[hd] : nullptr; H.dbeta = epi ? dbeta[hd] : nullptr;
    if (da && da[hd]) {
      if (sum_da) a.da = da[0]; else a.da2[hd] = da[hd];
    }
  }
  SUG_REQUIRE(!sum_da || !da || da[0], "sug_head_linear_bwd: the summed input gradient goes to da[0]");
  for (int hd = heads; hd < HMAX; ++hd) a.h[hd] = a.h[0];
  a.ldg = (int)ldg; a.ldin = (int)ldin; a.ldda = (int)ldda; a.M = M; a.K = K; a.No = No; a.NoP = (No + 31) / 32 * 32;
  a.heads = heads; a.sum_da = sum_da ? 1 : 0;
  a.slope = slope; a.eps = eps; a.p = p_drop; a.keep = 1.0f / (1.0f - p_drop);
  a.p_in = p_drop_in; a.keep_in = 1.0f / (1.0f - p_drop_in);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(K / 32, sum_da ? 1 : heads), block(NT);
  const size_t sh = (size_t)(32 * (a.NoP + 1) + 32 * 33 + NW * 32 * 32) * sizeof(float);
  static SugLdsOptIn note[12];
#define SUG_HEAD_BWD(EPI_, PRO_, NOP_, SLOT_)                                                                       \
  do {                                                                                                              \
    if (int rc = sug_allow_dynamic_lds(note[SLOT_], &head_bwd_kernel<EPI_, PRO_, NOP_>, 110 * 1024, "sug_head_linear_bwd")) return rc; \
    hipLaunchKernelGGL((head_bwd_kernel<EPI_, PRO_, NOP_>), grid, block, sh, st, a);                                \
  } while (0)
#define SUG_HEAD_BWD_N(EPI_, PRO_, SLOT_)                                                                           \
  do {                                                                                                              \
    if (a.NoP == 32) SUG_HEAD_BWD(EPI_, PRO_, 32, SLOT_);                                                           \
    else if (a.NoP == 256) SUG_HEAD_BWD(EPI_, PRO_, 256, SLOT_ + 1);                                                \
    else SUG_HEAD_BWD(EPI_, PRO_, 512, SLOT_ + 2);                                                                  \
  } while (0)
  if (epi && pro) SUG_HEAD_BWD_N(true, true, 0);
  else if (epi) SUG_HEAD_BWD_N(true, false, 3);
  else if (pro) SUG_HEAD_BWD_N(false, true, 6);
  else SUG_HEAD_BWD_N(false, false, 9);
#undef SUG_HEAD_BWD_N
#undef SUG_HEAD_BWD
  SUG_LAUNCH_CHECK("sug_head_linear_bwd");
  return SUG_OK;
}
