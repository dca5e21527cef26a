// Channel attention on the flattened SA-node features (CALayer, model/Model.py:16-34) for a handful of rows:
//     v [M, C=4096] -> h = relu(v . W0^T + b0) [M, 512] -> z = h . W2^T + b2 [M, 4096] -> BatchNorm1d(v * sigmoid(z) + v)
// (the two 1x1 Conv2d on a 1x1 map are GEMMs with M = the clouds of a domain, <= 64), for BOTH attention layers of
// Net_MDA (attention_s on the source rows, attention_t on the target rows of a paired batch) in one launch per stage.
// As library calls this was ~28 launches per step: four skinny GEMMs forward and eight backward at 13-26 us each (one
// to 32 workgroups streaming 8.4 MB of weights), bias sums, ReLU kernels, gradient adds -- 0.33 ms of a 5 ms step for
// 0.27 GFLOP.  The work is weight traffic: 2 x 16.8 MB read forward, the same again plus 2 x 16.8 MB of weight
// gradients written backward; every kernel here spreads it over 256 workgroups:
//   forward   ca_gemm_nt_kernel   h partials: 16 column tiles x 8 K-slices x 2 layers (each workgroup 64 KB of W0)
//             ca_h_kernel         h = relu(sum of the 8 partials + b0)
//             ca_gemm_nt_kernel   z: 128 column tiles x 2 layers (each 64 KB of W2), + b2
//             (sug_gate_bn_fwd: gate + BatchNorm1d, bnpool.hip)
//   backward  (sug_gate_bn_bwd -> dz and the gate's own input gradient)
//             ca_bwd2_kernel      per 32 rows of W2: dW2 = dz^T . h, db2, and the partial of dh = dz . W2 over these 32 rows
//             ca_dh_kernel        dh = (sum of the 128 partials) * [h > 0]
//             ca_bwd1_kernel      per 32 columns of W0: dW0 = dh^T . v, dv = dh . W0 + the gate's input gradient (the
//                                 gradient accumulation of v inside the kernel), db0
// fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// register r of the result holds D[8 (r >> 2) + (r & 3) + 4 (l >> 5)][l & 31]); every sum has a fixed order, no atomics.
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AMAX = 2;          // attention layers per launch
constexpr int NT = 512;          // threads per workgroup (8 waves)
constexpr int NW = NT / 64;
constexpr int TS = 36;           // floats per staging-tile row (32 + pad)
constexpr int HD = 512;          // hidden width of the attention MLP (channel / reduction = 4096 / 8)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------- forward GEMMs
struct GemmArgs {
  const float* in[AMAX];    // [M, Ktot] rows (row stride ldin)
  const float* W[AMAX];     // [No, Ktot]
  const float* bias[AMAX];  // [No] or null (ksplit == 1)
  float* out[AMAX];         // ksplit == 1: [M, No]; else partials [ksplit][M][No]
  int ldin, M, Ktot, No, ksplit;
};

// out[M, 32 columns] (+)= in[M, K slice] . W[32 rows, K slice]^T.  Workgroup = (column tile, K slice, layer); the slice is
// split over the 8 waves, a wave walks its part in sub-steps of 32 k: the 32 W rows and the M input rows of a sub-step are
// loaded with 8 lanes per row (whole 128-byte lines), written to the wave's own LDS tile and read back as MFMA operands
// (the scheme of heads.hip: "lane = row" loads straight from global memory run at the cache's tag rate).
template <int MB>                                                      // row blocks of 32 (M <= 32 MB)
__global__ __launch_bounds__(NT) void ca_gemm_nt_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TROWS = 32 + 32 * MB;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32, kz = blockIdx.y, la = blockIdx.z;
  const float* __restrict__ in = a.in[la];
  const float* __restrict__ W = a.W[la];
  const int M = a.M, Ktot = a.Ktot, No = a.No, ldin = a.ldin;
  const int Kwg = Ktot / a.ksplit, KW = Kwg / NW, nss = KW >> 5;
  const int kbase = kz * Kwg + w * KW;
  const int lr = lane >> 3, lp = lane & 7;
  float* tile = smem + w * TROWS * TS;
  f32x16 acc[MB];
#pragma unroll
  for (int rb = 0; rb < MB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  float4 wv[4], av[4 * MB];
  auto load = [&](int ss) {
    const int k = kbase + ss * 32 + lp * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[q] = ld4(W + (int64_t)min(n0 + q * 8 + lr, No - 1) * Ktot + k);
#pragma unroll
    for (int q = 0; q < 4 * MB; ++q) av[q] = ld4(in + (int64_t)min(q * 8 + lr, M - 1) * ldin + k);
  };
  load(0);
  for (int ss = 0; ss < nss; ++ss) {
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(tile + (q * 8 + lr) * TS + lp * 4, wv[q]);
#pragma unroll
    for (int q = 0; q < 4 * MB; ++q) st4(tile + (32 + q * 8 + lr) * TS + lp * 4, av[q]);
    if (ss + 1 < nss) load(ss + 1);
    __builtin_amdgcn_wave_barrier();
    const float* bp = tile + j * TS + 16 * h;                          // lane (j, h) takes k = 16 h + e of the sub-step
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = ld4(bp + 4 * g);
#pragma unroll
      for (int rb = 0; rb < MB; ++rb) {
        const float4 a4 = ld4(tile + (32 + rb * 32 + j) * TS + 16 * h + 4 * g);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[rb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the 8 partial tiles -> LDS (over the staging tiles) -> ordered sum (+ bias)
  __syncthreads();
  float* s_red = smem;                                                 // [NW][32 MB][32]
#pragma unroll
  for (int rb = 0; rb < MB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[(w * 32 * MB + rb * 32 + 8 * (r >> 2) + (r & 3) + 4 * h) * 32 + j] = acc[rb][r];
  __syncthreads();
  float* out = a.out[la] + (a.ksplit > 1 ? (int64_t)kz * M * No : 0);
  const float* bias = a.ksplit > 1 ? nullptr : a.bias[la];
  for (int e = t; e < 32 * MB * 32; e += NT) {
    const int i = e >> 5, jj = e & 31;
    float s = s_red[e];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) s += s_red[ww * 32 * MB * 32 + e];
    if (i < M && n0 + jj < No) out[(int64_t)i * No + n0 + jj] = s + (bias ? bias[n0 + jj] : 0.f);
  }
}

struct HArgs {
  const float* part[AMAX];  // [ksplit][M][HD]
  const float* bias[AMAX];
  float* h[AMAX];           // [M][HD]
  int M, ksplit;
};
__global__ __launch_bounds__(256) void ca_h_kernel(HArgs a) {
  const int la = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.M * HD) return;
  const float* p = a.part[la] + e;
  float s = p[0];
  for (int kz = 1; kz < a.ksplit; ++kz) s += p[(int64_t)kz * a.M * HD];
  s += a.bias[la][e % HD];
  a.h[la][e] = s > 0.f ? s : 0.f;
}

// ---------------------------------------------------------------------------------------------------- backward
struct Bwd2Args {
  const float* dz[AMAX];    // [M, C]
  const float* h[AMAX];     // [M, HD]
  const float* W2[AMAX];    // [C, HD]
  float* dW2[AMAX];         // [C, HD]
  float* db2[AMAX];         // [C]
  float* dhp[AMAX];         // [C / 32][M][HD] partials of dh
  int M, C;
};

// Workgroup = rows [n0, n0 + 32) of W2 (= columns of dz) of one layer.
//   dW2[n0 + i][c] = sum_m dz[m][n0 + i] h[m][c]      A = dz^T (LDS), B = h (global, one 128-byte segment per lane half)
//   dhp[m][c]      = sum_i dz[m][n0 + i] W2[n0 + i][c]   A = dz (LDS), B = W2 (global: this workgroup's 64 KB of the weight)
// wave w owns the columns [64 w, 64 w + 64) of both products.
template <int MB>
__global__ __launch_bounds__(NT) void ca_bwd2_kernel(Bwd2Args a) {
  __shared__ float s_dz[32 * MB][33];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, hh = lane >> 5;
  const int tile = blockIdx.x, n0 = tile * 32, la = blockIdx.y;
  const int M = a.M, C = a.C;
  const float* __restrict__ dz = a.dz[la];
  const float* __restrict__ h = a.h[la];
  const float* __restrict__ W2 = a.W2[la];
  for (int e = t; e < 32 * MB * 32; e += NT) {
    const int m = e >> 5, i = e & 31;
    s_dz[m][i] = m < M ? dz[(int64_t)m * C + n0 + i] : 0.f;
  }
  __syncthreads();
  if (t < 32) {
    float s = 0.f;
    for (int m = 0; m < 32 * MB; ++m) s += s_dz[m][t];                 // (rows >= M are zero)
    a.db2[la][n0 + t] = s;
  }
  const int c0 = w * 64;
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16 * MB; ++s) {
      const int m = 2 * s + hh;
      const float av = s_dz[m][j];
      const float* hr = h + (int64_t)min(m, M - 1) * HD + c0 + j;      // (dz rows >= M are zero)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hr[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hr[32], acc1, 0, 0, 0);
    }
    float* o = a.dW2[la] + (int64_t)n0 * HD + c0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + (r & 3) + 4 * hh;
      o[(int64_t)i * HD] = acc0[r];
      o[(int64_t)i * HD + 32] = acc1[r];
    }
  }
  {
    f32x16 acc[MB][2];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[rb][0][r] = 0.f; acc[rb][1][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int i = 2 * s + hh;
      const float* wr = W2 + (int64_t)(n0 + i) * HD + c0 + j;
      const float b0 = wr[0], b1 = wr[32];
#pragma unroll
      for (int rb = 0; rb < MB; ++rb) {
        const float av = s_dz[rb * 32 + j][i];
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[rb][1], 0, 0, 0);
      }
    }
    float* o = a.dhp[la] + (int64_t)tile * M * HD + c0 + j;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rb * 32 + 8 * (r >> 2) + (r & 3) + 4 * hh;
        if (m < M) {
          o[(int64_t)m * HD] = acc[rb][0][r];
          o[(int64_t)m * HD + 32] = acc[rb][1][r];
        }
      }
  }
}

struct DhArgs {
  const float* dhp[AMAX];   // [np][M][HD]
  const float* h[AMAX];
  float* dh[AMAX];          // [M][HD]: gradient of the pre-activation (ReLU applied)
  int M, np;
};
// workgroup = (32 columns, 8 rows, layer): 256 threads, thread = one element, the partials added in order
__global__ __launch_bounds__(256) void ca_dh_kernel(DhArgs a) {
  const int la = blockIdx.z;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), m = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (m >= a.M) return;
  const int64_t e = (int64_t)m * HD + c, stride = (int64_t)a.M * HD;
  const float* p = a.dhp[la] + e;
  float s = 0.f;
  int q = 0;
  for (; q + 8 <= a.np; q += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(q + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; q < a.np; ++q) s += p[q * stride];
  a.dh[la][e] = a.h[la][e] > 0.f ? s : 0.f;
}

struct Bwd1Args {
  const float* v[AMAX];     // [M, C] (row stride ldv): the layer input
  const float* dh[AMAX];    // [M, HD]
  const float* W0[AMAX];    // [HD, C]
  const float* dvg[AMAX];   // [M, C] the gate's own input gradient (added to dv), or null
  float* dW0[AMAX];         // [HD, C]
  float* db0[AMAX];         // [HD]
  float* dv[AMAX];          // [M, C] (row stride lddv)
  int M, C, ldv, lddv;
};

// Workgroup = columns [k0, k0 + 32) of W0 (= of v) of one layer; wave w owns the hidden units [64 w, 64 w + 64).
//   dW0[c][k0 + j] = sum_m dh[m][c] v[m][k0 + j]        A = dh^T (the wave's LDS slice), B = v (LDS)
//   dv[m][k0 + j]  = sum_c dh[m][c] W0[c][k0 + j]       A = dh (LDS slice), B = W0 (global: 128-byte segments of 512 rows),
//                    the 8 waves' partial tiles added in order, + the gate's input gradient
constexpr int DS = 65;           // row stride of a wave's dh slice (conflict-free column reads)
template <int MB>
__global__ __launch_bounds__(NT) void ca_bwd1_kernel(Bwd1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem;                                  // [32 MB][33]
  float* s_d = smem + 32 * MB * 33;                   // [NW][32 MB][DS]; afterwards s_red [NW][32 MB][32]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, hh = lane >> 5;
  const int k0 = blockIdx.x * 32, la = blockIdx.y;
  const int M = a.M, C = a.C;
  const float* __restrict__ v = a.v[la];
  const float* __restrict__ dh = a.dh[la];
  const float* __restrict__ W0 = a.W0[la];
  for (int e = t; e < 32 * MB * 32; e += NT) {
    const int m = e >> 5, i = e & 31;
    s_v[m * 33 + i] = m < M ? v[(int64_t)m * a.ldv + k0 + i] : 0.f;
  }
  float* sd = s_d + w * 32 * MB * DS;
  {
    // the wave's slice dh[:, 64 w .. 64 w + 64): 16 lanes per row (whole 256-byte pieces), 4 rows per instruction
    const int lr = lane >> 4, lp = lane & 15;
#pragma unroll
    for (int q = 0; q < 8 * MB; ++q) {
      const int m = q * 4 + lr;
      const float4 x = m < M ? ld4(dh + (int64_t)m * HD + w * 64 + lp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      float* d = sd + m * DS + lp * 4;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < HD) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dh[(int64_t)m * HD + t];
    a.db0[la][t] = s;
  }
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16 * MB; ++s) {
      const int m = 2 * s + hh;
      const float bv = s_v[m * 33 + j];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[m * DS + j], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[m * DS + 32 + j], bv, acc1, 0, 0, 0);
    }
    float* o = a.dW0[la] + (int64_t)(w * 64) * C + k0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + (r & 3) + 4 * hh;
      o[(int64_t)i * C] = acc0[r];
      o[(int64_t)(i + 32) * C] = acc1[r];
    }
  }
  f32x16 acc[MB];
#pragma unroll
  for (int rb = 0; rb < MB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll 8
  for (int s = 0; s < 32; ++s) {
    const int c = 2 * s + hh;
    const float bv = W0[(int64_t)(w * 64 + c) * C + k0 + j];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[(rb * 32 + j) * DS + c], bv, acc[rb], 0, 0, 0);
  }
  __syncthreads();                                    // every wave is done with its slice: the space becomes s_red
  float* s_red = s_d;
#pragma unroll
  for (int rb = 0; rb < MB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[(w * 32 * MB + rb * 32 + 8 * (r >> 2) + (r & 3) + 4 * hh) * 32 + j] = acc[rb][r];
  __syncthreads();
  const float* dvg = a.dvg[la];
  for (int e = t; e < 32 * MB * 32; e += NT) {
    const int m = e >> 5, jj = e & 31;
    float s = s_red[e];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) s += s_red[ww * 32 * MB * 32 + e];
    if (m < M) a.dv[la][(int64_t)m * a.lddv + k0 + jj] = s + (dvg ? dvg[(int64_t)m * C + k0 + jj] : 0.f);
  }
}

int ca_shape_ok(int A, int M, int C, int Hd) {
  return A >= 1 && A <= AMAX && M >= 1 && M <= 64 && Hd == HD && C >= 512 && C % 512 == 0 && C <= 8192;
}
}  // namespace

extern "C" int sug_calayer_supported(int A, int M, int C, int Hd) { return ca_shape_ok(A, M, C, Hd); }

extern "C" int sug_calayer_fwd(int A, const float* x, int64_t ldx, int M, int C, int Hd, const float* const* W0,
                               const float* const* b0, const float* const* W2, const float* const* b2, float* hp, float* h,
                               float* z, void* stream) {
  SUG_REQUIRE(ca_shape_ok(A, M, C, Hd), "sug_calayer_fwd: unsupported shape (layers %d, M %d, C %d, hidden %d)", A, M, C, Hd);
  SUG_REQUIRE(x && W0 && b0 && W2 && b2 && hp && h && z && ldx >= C && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0,
              "sug_calayer_fwd: bad operands");
  hipStream_t st = (hipStream_t)stream;
  const int ks = C / 512;
  GemmArgs g1, g2;
  HArgs ha;
  for (int la = 0; la < A; ++la) {
    SUG_REQUIRE(W0[la] && b0[la] && W2[la] && b2[la] && ((uintptr_t)W0[la] % 16) == 0 && ((uintptr_t)W2[la] % 16) == 0,
                "sug_calayer_fwd: null / unaligned weights of layer %d", la);
    g1.in[la] = x + (int64_t)la * M * ldx; g1.W[la] = W0[la]; g1.bias[la] = nullptr; g1.out[la] = hp + (int64_t)la * ks * M * HD;
    ha.part[la] = g1.out[la]; ha.bias[la] = b0[la]; ha.h[la] = h + (int64_t)la * M * HD;
    g2.in[la] = ha.h[la]; g2.W[la] = W2[la]; g2.bias[la] = b2[la]; g2.out[la] = z + (int64_t)la * M * C;
  }
  g1.ldin = (int)ldx; g1.M = M; g1.Ktot = C; g1.No = HD; g1.ksplit = ks;
  g2.ldin = HD; g2.M = M; g2.Ktot = HD; g2.No = C; g2.ksplit = 1;
  ha.M = M; ha.ksplit = ks;
  const int MB = M <= 32 ? 1 : 2;
  const size_t sh = (size_t)NW * (32 + 32 * MB) * TS * sizeof(float);
  static SugLdsOptIn note1, note2;
  if (MB == 1) {
    if (int rc = sug_allow_dynamic_lds(note1, &ca_gemm_nt_kernel<1>, (int)sh, "sug_calayer_fwd")) return rc;
    hipLaunchKernelGGL((ca_gemm_nt_kernel<1>), dim3(HD / 32, ks, A), dim3(NT), sh, st, g1);
    hipLaunchKernelGGL(ca_h_kernel, dim3(sug_divup((int64_t)M * HD, 256), A), dim3(256), 0, st, ha);
    hipLaunchKernelGGL((ca_gemm_nt_kernel<1>), dim3(C / 32, 1, A), dim3(NT), sh, st, g2);
  } else {
    if (int rc = sug_allow_dynamic_lds(note2, &ca_gemm_nt_kernel<2>, (int)sh, "sug_calayer_fwd")) return rc;
    hipLaunchKernelGGL((ca_gemm_nt_kernel<2>), dim3(HD / 32, ks, A), dim3(NT), sh, st, g1);
    hipLaunchKernelGGL(ca_h_kernel, dim3(sug_divup((int64_t)M * HD, 256), A), dim3(256), 0, st, ha);
    hipLaunchKernelGGL((ca_gemm_nt_kernel<2>), dim3(C / 32, 1, A), dim3(NT), sh, st, g2);
  }
  SUG_LAUNCH_CHECK("sug_calayer_fwd");
  return SUG_OK;
}

extern "C" int sug_calayer_bwd(int A, const float* x, int64_t ldx, int M, int C, int Hd, const float* const* W0,
                               const float* const* W2, const float* h, const float* dz, const float* dxg,
                               float* const* dW0, float* const* db0, float* const* dW2, float* const* db2, float* dhp,
                               float* dh, float* dx, int64_t lddx, void* stream) {
  SUG_REQUIRE(ca_shape_ok(A, M, C, Hd), "sug_calayer_bwd: unsupported shape (layers %d, M %d, C %d, hidden %d)", A, M, C, Hd);
  SUG_REQUIRE(x && W0 && W2 && h && dz && dW0 && db0 && dW2 && db2 && dhp && dh && dx && ldx >= C && lddx >= C,
              "sug_calayer_bwd: bad operands");
  hipStream_t st = (hipStream_t)stream;
  const int np = C / 32;
  Bwd2Args b2a;
  DhArgs da;
  Bwd1Args b1a;
  for (int la = 0; la < A; ++la) {
    SUG_REQUIRE(W0[la] && W2[la] && dW0[la] && db0[la] && dW2[la] && db2[la], "sug_calayer_bwd: null operand of layer %d", la);
    b2a.dz[la] = dz + (int64_t)la * M * C; b2a.h[la] = h + (int64_t)la * M * HD; b2a.W2[la] = W2[la];
    b2a.dW2[la] = dW2[la]; b2a.db2[la] = db2[la]; b2a.dhp[la] = dhp + (int64_t)la * np * M * HD;
    da.dhp[la] = b2a.dhp[la]; da.h[la] = b2a.h[la]; da.dh[la] = dh + (int64_t)la * M * HD;
    b1a.v[la] = x + (int64_t)la * M * ldx; b1a.dh[la] = da.dh[la]; b1a.W0[la] = W0[la];
    b1a.dvg[la] = dxg ? dxg + (int64_t)la * M * C : nullptr;
    b1a.dW0[la] = dW0[la]; b1a.db0[la] = db0[la]; b1a.dv[la] = dx + (int64_t)la * M * lddx;
  }
  b2a.M = M; b2a.C = C;
  da.M = M; da.np = np;
  b1a.M = M; b1a.C = C; b1a.ldv = (int)ldx; b1a.lddv = (int)lddx;
  const int MB = M <= 32 ? 1 : 2;
  const size_t sh1 = (size_t)(32 * MB * 33 + NW * 32 * MB * DS) * sizeof(float);
  static SugLdsOptIn n1, n2;
  if (MB == 1) {
    hipLaunchKernelGGL((ca_bwd2_kernel<1>), dim3(np, A), dim3(NT), 0, st, b2a);
    hipLaunchKernelGGL(ca_dh_kernel, dim3(HD / 32, sug_divup(M, 8), A), dim3(256), 0, st, da);
    if (int rc = sug_allow_dynamic_lds(n1, &ca_bwd1_kernel<1>, (int)sh1, "sug_calayer_bwd")) return rc;
    hipLaunchKernelGGL((ca_bwd1_kernel<1>), dim3(np, A), dim3(NT), sh1, st, b1a);
  } else {
    hipLaunchKernelGGL((ca_bwd2_kernel<2>), dim3(np, A), dim3(NT), 0, st, b2a);
    hipLaunchKernelGGL(ca_dh_kernel, dim3(HD / 32, sug_divup(M, 8), A), dim3(256), 0, st, da);
    if (int rc = sug_allow_dynamic_lds(n2, &ca_bwd1_kernel<2>, (int)sh1, "sug_calayer_bwd")) return rc;
    hipLaunchKernelGGL((ca_bwd1_kernel<2>), dim3(np, A), dim3(NT), sh1, st, b1a);
  }
  SUG_LAUNCH_CHECK("sug_calayer_bwd");
  return SUG_OK;
}
