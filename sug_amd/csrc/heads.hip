// Classifier heads (Pointnet_c, model/Model.py:412-449; fc_layer, model/model_utils.py:35-57) for a handful of rows:
//     x [M, K1] -> Linear -> LayerNorm -> act -> Dropout -> Linear -> LayerNorm -> act (= mid feature) -> Dropout -> Linear
// with M = the clouds of a step (<= 128) and both heads of Net_MDA (c1, c2: same input, own weights) in ONE launch per
// layer.  The library path is ~47 launches of 5-30 us per step for the two heads (three GEMMs with M = 64, bias sums,
// LayerNorm / activation / dropout kernels and their backwards, gradient adds) -- launch latency, not work: the whole
// arithmetic is 0.17 GFLOP and 5 MB of weights.  Here:
//   forward   3 launches  head_fwd_kernel:    z = pro(in) . W^T + b; pro = Dropout(act(LayerNorm(.))) of the PREVIOUS layer,
//             applied to the operand on its way into LDS (every workgroup forms the row statistics of its input rows itself);
//   backward  2 launches  head_ln_bwd_kernel: dz = LayerNorm / act / Dropout backward of a layer's output gradient, with the
//             LayerNorm weight gradients and the bias gradient (one workgroup per head: the rows are few and short);
//             3 launches  head_bwd_kernel:    dW = dz^T . pro(in) and the input gradient dz . W (+ the last layer's db).
// fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32); every sum has a fixed order (no atomics).
//
// Operand mapping of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// register r of the result holds D[i = 8 (r >> 2) + (r & 3) + 4 (l >> 5)][j = l & 31].  Which two k's a step multiplies
// is free as long as A and B agree.
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HMAX = 2;          // heads per launch
constexpr int NT = 512;          // threads per workgroup (8 waves)
constexpr int NW = NT / 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------- forward
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
  int ldin;
  int M, K, No;
  float slope, eps, p, keep;      // dropout: a value survives when u >= p and is scaled by keep = 1 / (1 - p)
};

// Workgroup = output columns [n0, n0 + 32) of one head, all rows (64 per pass).  The K range is split over the 8 waves;
// a wave walks its K/8 in sub-steps of 32 k (one 128-byte line per row): the 32 W rows and 64 input rows of a sub-step
// are loaded with 8 lanes per row (full lines -- "lane = row" loads straight into the MFMA operands touched 64 lines
// per instruction and ran at the cache's tag rate: 20 us for this 6 us kernel), written to the wave's OWN LDS tile
// [96][36] and read back as operands (lane = row).  No workgroup barrier inside the K loop.
constexpr int TS = 36;                 // floats per tile row (32 + pad)
constexpr int TROWS = 96;              // 32 W rows + 64 input rows
template <bool PRO>
__global__ __launch_bounds__(NT) void head_fwd_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_stats = smem;                       // [64][2]
  float* s_tile = smem + 128;                  // [NW][96][36]; after the K loop: s_red [NW][64][32]
  const FwdHead H = a.h[blockIdx.y];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  const int M = a.M, K = a.K, No = a.No, ldin = a.ldin;
  const int n0 = blockIdx.x * 32;
  const int lr = lane >> 3, lp = lane & 7;     // loads: 8 rows x 8 pieces of 16 bytes per instruction
  float* tile = s_tile + w * TROWS * TS;
  const int KW = K / NW, nss = KW >> 5;        // k per wave, sub-steps
  for (int r0 = 0; r0 < M; r0 += 64) {
    if constexpr (PRO) {
      // row statistics of `in` (two passes over registers, as torch's LayerNorm: mean, then the mean of squared
      // deviations); a wave takes 8 rows with all their loads in flight together (K <= 1024: 4 float4 per lane and row)
      __syncthreads();
      float4 v[8][4];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* row = H.in + min(r0 + w * 8 + r, M - 1) * ldin + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = (c * 256 < K) ? ld4(row + c * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
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
        const int i = w * 8 + r;
        if (lane == 0) {
          s_stats[2 * i] = mean; s_stats[2 * i + 1] = rstd;
          if (blockIdx.x == 0 && r0 + i < M && H.stats) { H.stats[2 * (r0 + i)] = mean; H.stats[2 * (r0 + i) + 1] = rstd; }
        }
      }
    }
    __syncthreads();                            // statistics visible; the previous pass's readers of s_red are done
    const bool write_act = PRO && blockIdx.x == 0 && H.act != nullptr;
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    float4 wv[4], av[8], uv[PRO ? 8 : 1], gv, bv;
    auto load = [&](int ss) {
      const int k = w * KW + ss * 32 + lp * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = ld4(H.W + min(n0 + q * 8 + lr, No - 1) * K + k);
#pragma unroll
      for (int q = 0; q < 8; ++q) av[q] = ld4(H.in + min(r0 + q * 8 + lr, M - 1) * ldin + k);
      if constexpr (PRO) {
        gv = ld4(H.gamma + k); bv = ld4(H.beta + k);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          uv[q] = H.u ? ld4(H.u + min(r0 + q * 8 + lr, M - 1) * K + k) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
    };
    load(0);
    for (int ss = 0; ss < nss; ++ss) {
      // registers -> the wave's tile (rows 0..31: W, 32..95: input), the LayerNorm / act / dropout of the input on the way
      const int k = w * KW + ss * 32 + lp * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) st4(tile + (q * 8 + lr) * TS + lp * 4, wv[q]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 v = av[q];
        if constexpr (PRO) {
          const int il = q * 8 + lr;
          const float mean = s_stats[2 * il], rstd = s_stats[2 * il + 1];
          float4 y;
          y.x = fmaf((v.x - mean) * rstd, gv.x, bv.x); y.y = fmaf((v.y - mean) * rstd, gv.y, bv.y);
          y.z = fmaf((v.z - mean) * rstd, gv.z, bv.z); y.w = fmaf((v.w - mean) * rstd, gv.w, bv.w);
          v.x = y.x > 0.f ? y.x : a.slope * y.x; v.y = y.y > 0.f ? y.y : a.slope * y.y;
          v.z = y.z > 0.f ? y.z : a.slope * y.z; v.w = y.w > 0.f ? y.w : a.slope * y.w;
          if (write_act && r0 + il < M) st4(H.act + (r0 + il) * K + k, v);
          if (H.u) {
            const float4 uu = uv[q];
            v.x = uu.x >= a.p ? v.x * a.keep : 0.f; v.y = uu.y >= a.p ? v.y * a.keep : 0.f;
            v.z = uu.z >= a.p ? v.z * a.keep : 0.f; v.w = uu.w >= a.p ? v.w * a.keep : 0.f;
          }
        }
        st4(tile + (32 + q * 8 + lr) * TS + lp * 4, v);
      }
      if (ss + 1 < nss) load(ss + 1);           // next sub-step's lines fly under this one's MFMAs
      __builtin_amdgcn_wave_barrier();          // (the tile is written and read by different lanes of this wave only)
      // operands: lane (j, h) takes k = 16 h + e of the sub-step
      const float* bp = tile + j * TS + 16 * h;
      const float* ap0 = tile + (32 + j) * TS + 16 * h;
      const float* ap1 = tile + (64 + j) * TS + 16 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = ld4(bp + 4 * g), a0 = ld4(ap0 + 4 * g), a1 = ld4(ap1 + 4 * g);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b4.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b4.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b4.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b4.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b4.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b4.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b4.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b4.w, acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // the 8 partial tiles -> LDS (over the staging tiles: every wave is done with its own) -> ordered sum + bias
    __syncthreads();
    float* s_red = s_tile;                      // [NW][64][32]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[(w * 64 + rb * 32 + 8 * (r >> 2) + (r & 3) + 4 * h) * 32 + j] = acc[rb][r];
    __syncthreads();
    for (int e = t; e < 64 * 32; e += NT) {
      const int i = e >> 5, jj = e & 31;
      float s = s_red[e];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) s += s_red[ww * 2048 + e];
      if (r0 + i < M && n0 + jj < No) H.z[(r0 + i) * No + n0 + jj] = s + (H.bias ? H.bias[n0 + jj] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------ LayerNorm / act / dropout backward
struct LnHead {
  const float* gup;      // [M, No] (row stride ldg): gradient of Dropout(act(LayerNorm(z)))
  const float* z;        // [M, No] pre-LayerNorm output, stats [M, 2], LayerNorm weights [No]
  const float* stats;
  const float* gamma;
  const float* beta;
  const float* u;        // [M, No] dropout randoms, or null
  const float* gextra;   // [M, No] extra gradient of the activation BEFORE the dropout (the mid feature's), or null
  float* dz;             // [M, No]
  float* dgamma;         // [No]
  float* dbeta;
  float* db;             // [No] column sums of dz, or null
};
struct LnArgs {
  LnHead h[HMAX];
  int ldg, M, No;
  float slope, p, keep;
};

// One workgroup (512 threads: a 256-register budget) per head: 16 rows at a time, 32 threads per row, NU float4 pieces
// per thread.
template <int NU>                                  // No = 128 NU (256 -> 2, 512 -> 4)
__global__ __launch_bounds__(NT) void head_ln_bwd_kernel(LnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [16][No]
  const LnHead H = a.h[blockIdx.x];
  const int t = threadIdx.x, i = t >> 5, q = t & 31;
  const int M = a.M, No = a.No;
  float4 pg[NU], pb[NU], pd[NU];                  // this thread's column partials over the row blocks: dgamma | dbeta | db
#pragma unroll
  for (int u = 0; u < NU; ++u) pg[u] = pb[u] = pd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = 0; r0 < M; r0 += 16) {
    const int row = r0 + i;
    const bool ok = row < M;
    const int rc = ok ? row : M - 1;
    const float mean = H.stats[2 * rc], rstd = H.stats[2 * rc + 1];
    float4 zz[NU], dd[NU], gg[NU];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = (q + 32 * u) * 4;
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
      float4 zc, g, d;
      zc.x = (z4.x - mean) * rstd; zc.y = (z4.y - mean) * rstd; zc.z = (z4.z - mean) * rstd; zc.w = (z4.w - mean) * rstd;
      g.x = fmaf(zc.x, g4.x, b4.x) > 0.f ? gu.x : a.slope * gu.x;
      g.y = fmaf(zc.y, g4.y, b4.y) > 0.f ? gu.y : a.slope * gu.y;
      g.z = fmaf(zc.z, g4.z, b4.z) > 0.f ? gu.z : a.slope * gu.z;
      g.w = fmaf(zc.w, g4.w, b4.w) > 0.f ? gu.w : a.slope * gu.w;
      if (!ok) g = make_float4(0.f, 0.f, 0.f, 0.f);
      d.x = g.x * g4.x; d.y = g.y * g4.y; d.z = g.z * g4.z; d.w = g.w * g4.w;
      zz[u] = zc; dd[u] = d; gg[u] = g;
      s1 += (d.x + d.y) + (d.z + d.w);
      s2 += (d.x * zc.x + d.y * zc.y) + (d.z * zc.z + d.w * zc.w);
    }
    // sums over the 32 threads of the row (xor tree inside a half wave: fixed order)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 / (float)No, m2 = s2 / (float)No;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = (q + 32 * u) * 4;
      float4 dz;
      dz.x = rstd * (dd[u].x - m1 - zz[u].x * m2); dz.y = rstd * (dd[u].y - m1 - zz[u].y * m2);
      dz.z = rstd * (dd[u].z - m1 - zz[u].z * m2); dz.w = rstd * (dd[u].w - m1 - zz[u].w * m2);
      if (ok) {
        st4(H.dz + row * No + c, dz);
        pd[u].x += dz.x; pd[u].y += dz.y; pd[u].z += dz.z; pd[u].w += dz.w;
      }
      pg[u].x += gg[u].x * zz[u].x; pg[u].y += gg[u].y * zz[u].y; pg[u].z += gg[u].z * zz[u].z; pg[u].w += gg[u].w * zz[u].w;
      pb[u].x += gg[u].x; pb[u].y += gg[u].y; pb[u].z += gg[u].z; pb[u].w += gg[u].w;
    }
  }
  // column sums over the 16 row threads: through LDS, rows added in ascending order
  auto fold = [&](const float4 (&part)[NU], float* out) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) st4(smem + i * No + (q + 32 * u) * 4, part[u]);
    __syncthreads();
    if (out) {
      for (int c = t; c < No; c += NT) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += smem[r * No + c];
        out[c] = s;
      }
    }
  };
  fold(pg, H.dgamma);
  fold(pb, H.dbeta);
  fold(pd, H.db);
}

// ---------------------------------------------------------------------------------------------------- backward GEMMs
struct BwdHead {
  const float* dz;       // [M, No] (row stride ldg): gradient of this layer's Linear output
  const float* in;       // [M, K] (row stride ldin): x, or (PRO) the previous layer's z with its statistics / LayerNorm / dropout
  const float* stats_in;
  const float* gamma_in;
  const float* beta_in;
  const float* u_in;
  const float* W;        // [No, K]
  float* dW;             // [No, K]
  float* db;             // [No] = column sums of dz, or null (the layers behind a LayerNorm get theirs from head_ln_bwd_kernel)
};
struct BwdArgs {
  BwdHead h[HMAX];
  float* da;             // sum_da: one [M, K] input gradient summed over the heads; else per head in da2
  float* da2[HMAX];
  int ldg, ldin, ldda;   // (row strides: 32-bit index arithmetic throughout -- every operand here is a few MB at most)
  int M, K, No, heads, sum_da;
  float slope, p_in, keep_in;
};

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
//   dW[n][k0 + j] = sum_i dz[i][n] a[i][k0 + j]   A = dz^T (lane = n), B = a (lane = column), k = the rows (32 per block)
//   da[i][k0 + j] = sum_n dz[i][n] W[n][k0 + j]   A = dz (lane = row), B = W (registers), n split over the waves
template <bool PRO, int NOP>
__global__ __launch_bounds__(NT) void head_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RS = NOP + 1;
  constexpr int MAXT = (NOP / 32 + NW - 1) / NW;       // dW tiles (32 output rows each) per wave
  const int M = a.M, K = a.K, No = a.No;
  float* s_dz = smem;                          // [32][NOP + 1]
  float* s_a = s_dz + 32 * RS;                 // [32][33]
  float* s_red = s_a + 32 * 33;                // [NW][32][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.x * 32;
  const int hd0 = a.sum_da ? 0 : blockIdx.y, hd1 = a.sum_da ? a.heads : blockIdx.y + 1;
  constexpr int ntile = NOP >> 5;              // dW tiles of 32 output rows; wave w owns tiles w, w + 8, ...
  constexpr int NWK = NOP / NW;                // output rows per wave in the dz . W product
  constexpr int nh = NWK >> 1;                 // ... per lane half
  const int nblk = (M + 31) >> 5;
  float dsum[2 * 4];                           // sum_da: this thread's 2 elements of every row block (<= 4), over the heads
#pragma unroll
  for (int e = 0; e < 8; ++e) dsum[e] = 0.f;

  for (int hd = hd0; hd < hd1; ++hd) {
    const BwdHead H = a.h[hd];
    // W[n][k0 + j] for this wave's n range, lane half h the contiguous half: registers (rows >= No meet zero dz columns)
    float wreg[nh];
#pragma unroll
    for (int s = 0; s < nh; ++s) wreg[s] = H.W[min(w * NWK + h * nh + s, No - 1) * K + k0 + j];
    f32x16 accw[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[q][r] = 0.f;
    float dbs = 0.f;                           // workgroup 0, thread = column: bias gradient

    for (int blk = 0; blk < nblk; ++blk) {
      const int r0 = blk * 32;
      __syncthreads();                          // the previous block's readers of s_dz / s_a / s_red are done
      // dz rows r0 .. r0+31 -> s_dz (columns >= No and rows >= M: zero)
      if constexpr (NOP % 128 == 0) {
        const int i = t >> 4, q = t & 15, row = r0 + i;
#pragma unroll
        for (int u = 0; u < NOP / 64; ++u) {
          const int c = (q + 16 * u) * 4;
          const float4 v = row < M ? ld4(H.dz + row * a.ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          float* d = s_dz + i * RS + c;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      } else {
        const int i = t >> 4, q = t & 15, row = r0 + i;
#pragma unroll
        for (int c0 = 0; c0 < NOP; c0 += 16) {
          const int c = c0 + q;
          s_dz[i * RS + c] = (row < M && c < No) ? H.dz[row * a.ldg + c] : 0.f;
        }
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
      if (H.db && blockIdx.x == 0 && t < No) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) dbs += s_dz[i * RS + t];       // (rows >= M are zero)
      }
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
    if (H.db && blockIdx.x == 0 && t < No) H.db[t] = dbs;
  }
}

int heads_ok(int heads) { return heads >= 1 && heads <= HMAX; }

}  // namespace

extern "C" int sug_head_linear_supported(int M, int K, int No, int pro, int epi) {
  if (M < 1 || M > 128 || K < 256 || K % 256 || K > 4096 || No < 1) return 0;     // (K / 8 per wave in 32-k sub-steps)
  if (!(No <= 32 || No == 256 || No == 512)) return 0;   // (the backward kernels are instantiated for these output widths)
  if (pro && K > 1024) return 0;                         // (row statistics pass: 4 float4 per lane and row)
  if (epi && !(No == 256 || No == 512)) return 0;
  return 1;
}

extern "C" int sug_head_linear_fwd(int heads, const float* const* in, int64_t ldin, const float* const* W,
                                   const float* const* bias, float* const* z, const float* const* gamma,
                                   const float* const* beta, const float* const* u, float* const* stats,
                                   float* const* act, int M, int K, int No, int pro, float slope, float eps, float p_drop,
                                   void* stream) {
  SUG_REQUIRE(heads_ok(heads), "sug_head_linear_fwd: 1 or 2 heads per launch, got %d", heads);
  SUG_REQUIRE(sug_head_linear_supported(M, K, No, pro, 0), "sug_head_linear_fwd: unsupported shape M=%d K=%d No=%d", M, K, No);
  SUG_REQUIRE(in && W && z && ldin >= K && ldin % 4 == 0 && ldin < (1 << 20), "sug_head_linear_fwd: bad operands");
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
  a.ldin = (int)ldin; a.M = M; a.K = K; a.No = No; a.slope = slope; a.eps = eps; a.p = p_drop; a.keep = 1.0f / (1.0f - p_drop);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((No + 31) / 32, heads), block(NT);
  const size_t sh = (size_t)(128 + NW * TROWS * TS) * sizeof(float);         // 108.5 KB (the reduction's 64 KB alias the tiles)
  static SugLdsOptIn note[2];
  if (pro) {
    if (int rc = sug_allow_dynamic_lds(note[0], &head_fwd_kernel<true>, 112 * 1024, "sug_head_linear_fwd")) return rc;
    hipLaunchKernelGGL((head_fwd_kernel<true>), grid, block, sh, st, a);
  } else {
    if (int rc = sug_allow_dynamic_lds(note[1], &head_fwd_kernel<false>, 112 * 1024, "sug_head_linear_fwd")) return rc;
    hipLaunchKernelGGL((head_fwd_kernel<false>), grid, block, sh, st, a);
  }
  SUG_LAUNCH_CHECK("sug_head_linear_fwd");
  return SUG_OK;
}

extern "C" int sug_head_ln_bwd(int heads, const float* const* gup, int64_t ldg, const float* const* z,
                               const float* const* stats, const float* const* gamma, const float* const* beta,
                               const float* const* u, const float* const* gextra, float* const* dz, float* const* dgamma,
                               float* const* dbeta, float* const* db, int M, int No, float slope, float p_drop, void* stream) {
  SUG_REQUIRE(heads_ok(heads), "sug_head_ln_bwd: 1 or 2 heads per launch, got %d", heads);
  SUG_REQUIRE(M >= 1 && M <= 128 && (No == 256 || No == 512), "sug_head_ln_bwd: unsupported shape M=%d No=%d", M, No);
  SUG_REQUIRE(gup && z && stats && gamma && beta && dz && dgamma && dbeta && ldg >= No && ldg % 4 == 0 && ldg < (1 << 20),
              "sug_head_ln_bwd: bad operands");
  SUG_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sug_head_ln_bwd: dropout probability %g", (double)p_drop);
  LnArgs a;
  for (int hd = 0; hd < heads; ++hd) {
    SUG_REQUIRE(gup[hd] && z[hd] && stats[hd] && gamma[hd] && beta[hd] && dz[hd] && dgamma[hd] && dbeta[hd],
                "sug_head_ln_bwd: null operand of head %d", hd);
    LnHead& H = a.h[hd];
    H.gup = gup[hd]; H.z = z[hd]; H.stats = stats[hd]; H.gamma = gamma[hd]; H.beta = beta[hd];
    H.u = u ? u[hd] : nullptr; H.gextra = gextra ? gextra[hd] : nullptr;
    H.dz = dz[hd]; H.dgamma = dgamma[hd]; H.dbeta = dbeta[hd]; H.db = db ? db[hd] : nullptr;
  }
  for (int hd = heads; hd < HMAX; ++hd) a.h[hd] = a.h[0];
  a.ldg = (int)ldg; a.M = M; a.No = No; a.slope = slope; a.p = p_drop; a.keep = 1.0f / (1.0f - p_drop);
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)16 * No * sizeof(float);            // <= 32 KB
  if (No == 256) hipLaunchKernelGGL((head_ln_bwd_kernel<2>), dim3(heads), dim3(NT), sh, st, a);
  else hipLaunchKernelGGL((head_ln_bwd_kernel<4>), dim3(heads), dim3(NT), sh, st, a);
  SUG_LAUNCH_CHECK("sug_head_ln_bwd");
  return SUG_OK;
}

extern "C" int sug_head_linear_bwd(int heads, int sum_da, const float* const* dz, int64_t ldg, const float* const* in,
                                   int64_t ldin, const float* const* stats_in, const float* const* gamma_in,
                                   const float* const* beta_in, const float* const* u_in, const float* const* W,
                                   float* const* dW, float* const* db, float* const* da, int64_t ldda, int M, int K, int No,
                                   int pro, float slope, float p_drop_in, void* stream) {
  SUG_REQUIRE(heads_ok(heads), "sug_head_linear_bwd: 1 or 2 heads per launch, got %d", heads);
  SUG_REQUIRE(sug_head_linear_supported(M, K, No, pro, 0), "sug_head_linear_bwd: unsupported shape M=%d K=%d No=%d", M, K, No);
  SUG_REQUIRE(dz && in && W && dW && ldg >= No && ldin >= K && ldg < (1 << 20) && ldin < (1 << 20) && ldda < (1 << 20),
              "sug_head_linear_bwd: bad operands");
  SUG_REQUIRE(No <= 32 || ldg % 4 == 0, "sug_head_linear_bwd: gradient rows must be 16-byte aligned");
  SUG_REQUIRE(p_drop_in >= 0.f && p_drop_in < 1.f, "sug_head_linear_bwd: dropout probability %g", (double)p_drop_in);
  BwdArgs a;
  a.da = nullptr;
  for (int hd = 0; hd < HMAX; ++hd) a.da2[hd] = nullptr;
  for (int hd = 0; hd < heads; ++hd) {
    SUG_REQUIRE(dz[hd] && in[hd] && W[hd] && dW[hd], "sug_head_linear_bwd: null operand of head %d", hd);
    SUG_REQUIRE(!pro || (stats_in && gamma_in && beta_in && stats_in[hd] && gamma_in[hd] && beta_in[hd]),
                "sug_head_linear_bwd: LayerNorm operands of the input missing");
    SUG_REQUIRE(No <= 32 || ((uintptr_t)dz[hd] % 16) == 0, "sug_head_linear_bwd: dz must be 16-byte aligned");
    BwdHead& H = a.h[hd];
    H.dz = dz[hd];
    H.in = in[hd];
    H.stats_in = pro ? stats_in[hd] : nullptr; H.gamma_in = pro ? gamma_in[hd] : nullptr; H.beta_in = pro ? beta_in[hd] : nullptr;
    H.u_in = (pro && u_in) ? u_in[hd] : nullptr;
    H.W = W[hd]; H.dW = dW[hd]; H.db = db ? db[hd] : nullptr;
    if (da && da[hd]) {
      if (sum_da) a.da = da[0]; else a.da2[hd] = da[hd];
    }
  }
  SUG_REQUIRE(!sum_da || !da || da[0], "sug_head_linear_bwd: the summed input gradient goes to da[0]");
  for (int hd = heads; hd < HMAX; ++hd) a.h[hd] = a.h[0];
  a.ldg = (int)ldg; a.ldin = (int)ldin; a.ldda = (int)ldda; a.M = M; a.K = K; a.No = No;
  a.heads = heads; a.sum_da = sum_da ? 1 : 0;
  a.slope = slope; a.p_in = p_drop_in; a.keep_in = 1.0f / (1.0f - p_drop_in);
  const int NoP = (No + 31) / 32 * 32;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(K / 32, sum_da ? 1 : heads), block(NT);
  const size_t sh = (size_t)(32 * (NoP + 1) + 32 * 33 + NW * 32 * 32) * sizeof(float);
  static SugLdsOptIn note[6];
#define SUG_HEAD_BWD(PRO_, NOP_, SLOT_)                                                                             \
  do {                                                                                                              \
    if (int rc = sug_allow_dynamic_lds(note[SLOT_], &head_bwd_kernel<PRO_, NOP_>, 110 * 1024, "sug_head_linear_bwd")) return rc; \
    hipLaunchKernelGGL((head_bwd_kernel<PRO_, NOP_>), grid, block, sh, st, a);                                      \
  } while (0)
  if (pro) {
    if (NoP == 32) SUG_HEAD_BWD(true, 32, 0); else if (NoP == 256) SUG_HEAD_BWD(true, 256, 1); else SUG_HEAD_BWD(true, 512, 2);
  } else {
    if (NoP == 32) SUG_HEAD_BWD(false, 32, 3); else if (NoP == 256) SUG_HEAD_BWD(false, 256, 4); else SUG_HEAD_BWD(false, 512, 5);
  }
#undef SUG_HEAD_BWD
  SUG_LAUNCH_CHECK("sug_head_linear_bwd");
  return SUG_OK;
}
