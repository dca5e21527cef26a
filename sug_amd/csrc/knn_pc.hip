// kNN graph, producer / consumer formulation (C = 3, 64, 128, k <= 20): the hot index kernel of the
// DGCNN path.  Replaces knn(), model/model_utils.py:178-185.
//
// The score tiles <x_i, x_j> run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: bit for bit the
// ascending-k fmaf chain of the scalar kernel and of the CPU reference's K=3 sgemm), the top-k
// selection is VALU work.  A wave issues in order, so one wave cannot keep both pipes busy; here
// the two kinds of work live in DIFFERENT waves of one 512-thread workgroup and every SIMD hosts
// one of each:
//   waves 0-3  producers: stream 32-candidate tiles global -> registers -> LDS (mfma_tile.h), run
//              the MFMA chains of 64 queries each (two 32-query column blocks sharing the A
//              operand) and write the 64 x 32 score tile to LDS, row = query;
//   waves 4-7  consumers: lane = ONE query; reads its row of the score tile (candidates in ascending
//              index order), forms score = (-|x_j|^2 - (-2<x_i,x_j>)) - |x_i|^2 and compares with
//              thr = its K-th best score as of the previous tile; the few candidates that pass are
//              marked in a 32-bit mask and then inserted -- all lanes of the wave in lockstep, one
//              marked candidate per lane per iteration -- into the lane's sorted (score, index) lists
//              in registers (v_med3 for the scores, two v_cndmask for the indices per slot).
// One barrier per tile hands tile t's scores to the consumers while the producers work on tile
// t+1.  No candidate ring, no compaction, no merge: ties keep the lower index because candidates
// arrive in ascending order and insertion is strict; the lists are the result.
// LDS: tiles 3 x 32 x (C+4) + score tiles 2 x 4 x 64 x 36 floats = 100 / 125 KB (C = 64 / 128).  Workgroup = 256 queries of one cloud; the 4 workgroups of a 1024-point cloud
// share an XCD (one L2).
//
// FLOPs N^2*(2C+3) per cloud on the matrix pipe (157 TFLOP/s); algorithmic bytes 4*C*N + 4*N*k.
#include "common.h"
#include "mfma_tile.h"

namespace {
using namespace sug_tile;

constexpr int SROW = 36;        // floats per query row of a score tile (32 + pad: conflict-free b128)


__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// Sorted insertion of (s, j) into the descending lists v[] / id[], in place, 4 VALU per slot:
//   up = s > v[u-1];  id[u] = up ? id[u-1] : (here ? j : id[u]);  v[u] = med3(v[u-1], v[u], s);  here = up
// (hipcc's register allocation of the same loop in C++ copies all 40 list registers every iteration.)
// The "here" mask alternates between an SGPR pair `m` and VCC; gfx950 needs two wait states between a
// VALU write of an SGPR / VCC and a VALU read of it as a mask: each v_cndmask reads a mask written
// at least three instructions earlier.  Strict compares: an equal score stays behind earlier entries.
__device__ __forceinline__ void ins_first(unsigned long long& m, float s, float vlast) {
  asm volatile("v_cmp_gt_f32_e64 %[m], %[s], %[v]\n\ts_nop 1" : [m] "=s"(m) : [s] "v"(s), [v] "v"(vlast));
}
// slots u, u-1, u-2, u-3 (v4 = v[u] ... v0 = v[u-4], read only); here: in m, out m
__device__ __forceinline__ void ins_four(unsigned long long& m, float s, int j, float& v4, float& v3, float& v2,
                                         float& v1, const float& v0, int& i4, int& i3, int& i2, int& i1, const int& i0) {
  int t;
  asm volatile(
      "v_cmp_gt_f32_e32 vcc, %[s], %[v3]\n\t"
      "v_cndmask_b32_e64 %[t], %[i4], %[j], %[m]\n\t"
      "v_med3_f32 %[v4], %[v3], %[v4], %[s]\n\t"
      "v_cndmask_b32_e32 %[i4], %[t], %[i3], vcc\n\t"
      "v_cmp_gt_f32_e64 %[m], %[s], %[v2]\n\t"
      "v_cndmask_b32_e32 %[t], %[i3], %[j], vcc\n\t"
      "v_med3_f32 %[v3], %[v2], %[v3], %[s]\n\t"
      "v_cndmask_b32_e64 %[i3], %[t], %[i2], %[m]\n\t"
      "v_cmp_gt_f32_e32 vcc, %[s], %[v1]\n\t"
      "v_cndmask_b32_e64 %[t], %[i2], %[j], %[m]\n\t"
      "v_med3_f32 %[v2], %[v1], %[v2], %[s]\n\t"
      "v_cndmask_b32_e32 %[i2], %[t], %[i1], vcc\n\t"
      "v_cmp_gt_f32_e64 %[m], %[s], %[v0]\n\t"
      "v_cndmask_b32_e32 %[t], %[i1], %[j], vcc\n\t"
      "v_med3_f32 %[v1], %[v0], %[v1], %[s]\n\t"
      "v_cndmask_b32_e64 %[i1], %[t], %[i0], %[m]"
      : [m] "+s"(m), [t] "=&v"(t), [v4] "+v"(v4), [v3] "+v"(v3), [v2] "+v"(v2), [v1] "+v"(v1), [i4] "+v"(i4),
        [i3] "+v"(i3), [i2] "+v"(i2), [i1] "+v"(i1)
      : [s] "v"(s), [j] "v"(j), [v0] "v"(v0), [i0] "v"(i0)
      : "vcc");
}
// slots 3, 2, 1 and the head slot 0; here: in m
__device__ __forceinline__ void ins_tail(unsigned long long m, float s, int j, float& v3, float& v2, float& v1, float& v0,
                                         int& i3, int& i2, int& i1, int& i0) {
  int t;
  asm volatile(
      "v_cmp_gt_f32_e32 vcc, %[s], %[v2]\n\t"
      "v_cndmask_b32_e64 %[t], %[i3], %[j], %[m]\n\t"
      "v_med3_f32 %[v3], %[v2], %[v3], %[s]\n\t"
      "v_cndmask_b32_e32 %[i3], %[t], %[i2], vcc\n\t"
      "v_cmp_gt_f32_e64 %[m], %[s], %[v1]\n\t"
      "v_cndmask_b32_e32 %[t], %[i2], %[j], vcc\n\t"
      "v_med3_f32 %[v2], %[v1], %[v2], %[s]\n\t"
      "v_cndmask_b32_e64 %[i2], %[t], %[i1], %[m]\n\t"
      "v_cmp_gt_f32_e32 vcc, %[s], %[v0]\n\t"
      "v_cndmask_b32_e64 %[t], %[i1], %[j], %[m]\n\t"
      "v_med3_f32 %[v1], %[v0], %[v1], %[s]\n\t"
      "v_cndmask_b32_e32 %[i1], %[t], %[i0], vcc\n\t"
      "v_max_f32_e32 %[v0], %[v0], %[s]\n\t"
      "v_cndmask_b32_e32 %[i0], %[i0], %[j], vcc"
      : [m] "+s"(m), [t] "=&v"(t), [v3] "+v"(v3), [v2] "+v"(v2), [v1] "+v"(v1), [v0] "+v"(v0), [i3] "+v"(i3),
        [i2] "+v"(i2), [i1] "+v"(i1), [i0] "+v"(i0)
      : [s] "v"(s), [j] "v"(j)
      : "vcc");
}

template <int K>
__device__ __forceinline__ void insert_sorted(float (&v)[K], int (&id)[K], float s, int j) {
  static_assert(K % 4 == 0 && K >= 8, "list length: a multiple of 4");
  unsigned long long m;
  ins_first(m, s, v[K - 1]);
#pragma unroll
  for (int u = K - 1; u >= 7; u -= 4)
    ins_four(m, s, j, v[u], v[u - 1], v[u - 2], v[u - 3], v[u - 4], id[u], id[u - 1], id[u - 2], id[u - 3], id[u - 4]);
  ins_tail(m, s, j, v[3], v[2], v[1], v[0], id[3], id[2], id[1], id[0]);
}

template <int CP, int K>
__global__ __launch_bounds__(512, 2) void knn_pc_kernel(const float* __restrict__ x, int64_t ldx, int B, int N,
                                                        int k, int32_t* __restrict__ idx) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);                       // [3][TJ][RS]
  float* s_norm = s_tile + 3 * TJ * RS;                                 // [3][TJ] (+pad)
  float* s_score = s_norm + 4 * TJ;                                     // [2][4][64][SROW]

  const int nq = (N + 255) / 256;
  int b, qb;
  if ((B & 7) == 0) {                        // a cloud's query blocks share an XCD (its L2 holds the cloud once)
    const int grp = blockIdx.x / (8 * nq), rem = blockIdx.x % (8 * nq);
    b = grp * 8 + (rem & 7);
    qb = rem >> 3;
  } else {
    b = blockIdx.x / nq;
    qb = blockIdx.x % nq;
  }
  const float* xb = x + (int64_t)b * N * ldx;
  const bool producer = threadIdx.x < 256;
  const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;       // wv: producer / consumer pair index
  const int qj = lane & 31, h = lane >> 5;
  const int q0 = qb * 256;

  const int ntile = (N + TJ - 1) / TJ;
  auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };
  auto nbuf = [&](int t) { return s_norm + (t % 3) * TJ; };

  // The two roles are separate code paths (disjoint register sets); both execute the same sequence
  // of workgroup barriers: 17 in the query staging, 2 in the pipeline fill, one per tile.
  if (producer) {
    // ---- query operands, staged through the tile buffers: 8 tiles of 32 query rows
    float bq0[HALF], bq1[HALF];
    {
      TileRegs<CP> tr;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        __syncthreads();
        tile_load<CP>(tr, xb, ldx, N, q0 + w * TJ);
        tile_store<CP, true>(tr, s_tile, s_norm, N, q0 + w * TJ);
        __syncthreads();
        if ((w >> 1) == wv) {
          const float* qrow = s_tile + qj * RS + h * HALF;
          if (w & 1) {
#pragma unroll
            for (int e = 0; e < HALF; ++e) bq1[e] = qrow[e];
          } else {
#pragma unroll
            for (int e = 0; e < HALF; ++e) bq0[e] = qrow[e];
          }
        }
      }
    }
    __syncthreads();

    // scores of candidate tile t for this wave's 64 queries -> score buffer `buf`
    auto produce = [&](int t, int buf) {
      const float* arow = tbuf(t) + qj * RS + h * HALF;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      // the two column blocks share the A operand; their chains alternate on the matrix pipe
      if constexpr (CP == 4) {
        const float2 a2 = *reinterpret_cast<const float2*>(arow);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, bq0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, bq1[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, bq0[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, bq1[1], acc1, 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < HALF / 4; ++g) {
          const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq0[4 * g + 0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq1[4 * g + 0], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq0[4 * g + 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq1[4 * g + 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq0[4 * g + 2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq1[4 * g + 2], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq0[4 * g + 3], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq1[4 * g + 3], acc1, 0, 0, 0);
        }
      }
      // S^T tile: lane = query column, registers 4g..4g+3 = candidate rows 8g + 4h .. +3: one b128 per g
      float* d0 = s_score + ((buf * 4 + wv) * 64 + qj) * SROW + 4 * h;
      float* d1 = d0 + 32 * SROW;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(d0 + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(d1 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
    };

    // pipeline: iteration t = scores of tile t+1 (consumers are on tile t); registers of tile t+2 -> LDS;
    // the global loads of tile t+3 are in flight for a whole iteration
    TileRegs<CP> tr;
    tile_load<CP>(tr, xb, ldx, N, 0);
    tile_store<CP, true>(tr, tbuf(0), nbuf(0), N, 0);
    if (ntile > 1) tile_load<CP>(tr, xb, ldx, N, TJ);
    __syncthreads();
    produce(0, 0);
    if (ntile > 1) tile_store<CP, true>(tr, tbuf(1), nbuf(1), N, TJ);
    if (ntile > 2) tile_load<CP>(tr, xb, ldx, N, 2 * TJ);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
      if (t + 1 < ntile) produce(t + 1, (t + 1) & 1);
      if (t + 2 < ntile) tile_store<CP, true>(tr, tbuf(t + 2), nbuf(t + 2), N, (t + 2) * TJ);
      if (t + 3 < ntile) tile_load<CP>(tr, xb, ldx, N, (t + 3) * TJ);
      __syncthreads();
    }
  } else {
    // ---- consumer: lane = one query; |x_i|^2 from the staged query tiles
    float ni = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      __syncthreads();
      __syncthreads();
      if ((w >> 1) == wv && h == (w & 1)) ni = s_norm[qj];
    }
    __syncthreads();

    float v[K];
    int id[K];
    const int q = q0 + wv * 64 + lane;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      v[t] = -INFINITY;
      id[t] = q < N ? q : 0;                   // never-filled slots (NaN features, N < k) point at the query
    }
    float thr = -INFINITY;

    // score of candidate slot c of the current tile: pairwise_distance = -xx - inner - xx^T, inner = -2*dot
    // (model_utils.py:179-181); the same three roundings wherever it is evaluated
    auto score = [&](float dot, float nj) { return __fsub_rn(__fsub_rn(-nj, __fmul_rn(-2.0f, dot)), ni); };

    __syncthreads();                            // pipeline fill: tile 0 staged
    __syncthreads();                            // tile 0 scored
    for (int t = 0; t < ntile; ++t) {
      const float* srow = s_score + (((t & 1) * 4 + wv) * 64 + lane) * SROW;
      const float* nrm = nbuf(t);
      // pass 1: which of the 32 candidates beat thr (the K-th best score as of the previous tile: stale
      // by at most one tile, never too high).  Candidate c -> bit 31-c.
      unsigned int mask = 0u;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float4 s4 = *reinterpret_cast<const float4*>(srow + 4 * g);
        const float4 n4 = *reinterpret_cast<const float4*>(nrm + 4 * g);
        mask = mask + mask + (score(s4.x, n4.x) > thr ? 1u : 0u);
        mask = mask + mask + (score(s4.y, n4.y) > thr ? 1u : 0u);
        mask = mask + mask + (score(s4.z, n4.z) > thr ? 1u : 0u);
        mask = mask + mask + (score(s4.w, n4.w) > thr ? 1u : 0u);
      }
      // pass 2: the marked candidates in ascending index order, one per lane per iteration (all lanes in
      // lockstep); the entry of the NEXT iteration is fetched from the score tile while the current
      // one is inserted.  Exact sorted insertion: v_med3 on the scores, two v_cndmask on the indices
      // per slot; strict compares keep the earlier (lower) index ahead among equal scores.
      // (loads are unconditional -- an unmarked lane re-reads slot 0 -- so that no branch wraps them;
      //  the entry of the next iteration is fetched while the current one is inserted)
      // iterations = the largest number of marked candidates of any lane: a wave-wide maximum built bit by
      // bit from ballots (scalar unit only; a shuffle tree would cost six LDS-latency round trips per tile)
      const int pc = __popc(mask);
      unsigned long long cand = ~0ull;
      int nit = 0;
#pragma unroll
      for (int bit = 5; bit >= 0; --bit) {
        const unsigned long long m = __ballot((pc >> bit) & 1) & cand;
        if (m) {
          nit |= 1 << bit;
          cand = m;
        }
      }
      // The LDS reads are issued by hand and waited for only after the insertion of the previous entry
      // (hipcc waits right at the load, or wraps the loads in a branch and then copies all 40 list
      // registers every iteration).
      const unsigned srow_a = lds_addr(srow), nrm_a = lds_addr(nrm);
      float dot, nj;
      bool valid;
      int c;
      auto fetch_issue = [&]() {
        valid = mask != 0u;
        c = valid ? __builtin_clz(mask) : 0;
        mask &= ~(0x80000000u >> c);
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(dot), "=&v"(nj) : "v"(srow_a + 4u * c), "v"(nrm_a + 4u * c));
      };
      auto fetch_finish = [&](float& s_out, int& j_out) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dot), "+v"(nj));
        s_out = valid ? score(dot, nj) : -INFINITY;
        j_out = t * TJ + c;
      };
      float s_cur;
      int j_cur;
      fetch_issue();
      fetch_finish(s_cur, j_cur);
      for (int it = 0; it < nit; ++it) {
        fetch_issue();
#ifdef SUG_KNN_EXPERIMENT
        {   // timing experiment only (wrong indices): one v_med3 per slot
#pragma unroll
          for (int u = K - 1; u >= 1; --u) asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(v[u]) : "v"(v[u - 1]), "v"(s_cur));
          asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[0]) : "v"(s_cur));
          id[0] += j_cur;
        }
#else
        insert_sorted<K>(v, id, s_cur, j_cur);
#endif
        fetch_finish(s_cur, j_cur);
      }
      thr = v[K - 1];
      __syncthreads();
    }
    if (q < N) {
      int32_t* o = idx + ((int64_t)b * N + q) * k;
#pragma unroll
      for (int t = 0; t < K; ++t)
        if (t < k) o[t] = id[t];
    }
  }
}

template <int CP, int K>
int launch_pc(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  constexpr int RS = CP + 4;
  const size_t sh = (size_t)(3 * TJ * RS + 4 * TJ + 2 * 4 * 64 * SROW) * sizeof(float);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_pc_kernel<CP, K>, (int)sh, "sug_knn(mfma, producer/consumer)")) return rc;
  dim3 grid(sug_divup(N, 256) * B);
  hipLaunchKernelGGL((knn_pc_kernel<CP, K>), grid, dim3(512), sh, st, x, ldx, B, N, k, idx);
  SUG_LAUNCH_CHECK("sug_knn(mfma, producer/consumer)");
  return SUG_OK;
}

template <int K>
int dispatch_pc(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st) {
  if (C == 3) return launch_pc<4, K>(x, ldx, B, N, k, idx, st);
  if (C == 64) return launch_pc<64, K>(x, ldx, B, N, k, idx, st);
  return launch_pc<128, K>(x, ldx, B, N, k, idx, st);
}

}  // namespace

int sug_knn_pc(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st) {
  if (k <= 16) return dispatch_pc<16>(x, ldx, B, N, C, k, idx, st);
  return dispatch_pc<20>(x, ldx, B, N, C, k, idx, st);
}
