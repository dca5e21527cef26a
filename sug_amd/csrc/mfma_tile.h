// Row-tile staging for the fp32 MFMA kernels (knn_pc.hip, pointmlp.hip): 32 rows of a
// [rows, CP] fp32 matrix travel global -> registers -> LDS, de-interleaved so that one
// ds_read_b128 feeds the A operand of four consecutive v_mfma_f32_32x32x2_f32 k-steps.
//
// LDS image of a tile: row r at s_tile + r*(CP+4); inside a row [even features | odd features]
// (CP/2 floats each).  Lane l of a wave (row i = l&31, half h = l>>5) reads its operands for
// k-steps s = 0..CP/2-1 from  s_tile + i*RS + h*CP/2 + s : the value is feature 2s+h, which is
// A[i][k = l>>5] of step s.  Row stride CP+4 floats keeps the b128 reads conflict-free.
#pragma once
#include "common.h"

namespace sug_tile {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TJ = 32;   // rows per tile = MFMA M

// Global -> register half of the staging (so the loads fly under the previous tile's MFMAs).
// NT: threads that stage a tile together (256, or 128 for the two-wave producers of knn_pc_kernel's 128-query form)
template <int CP, int NT = 256>
struct TileRegs {
  static constexpr int NV = (CP == 4) ? 1 : (4 * CP / NT);   // (row, 8-feature chunk) items per thread
  float4 lo[NV], hi[NV];
};

// rows [row0, row0+32) of xb (row stride ldx); rows >= N read row N-1 (their norm is stored as +inf by tile_store: such a
// candidate never scores, and the scores of such a query are never used).  NT threads, numbered by `tid` (0 .. NT-1).
template <int CP, int NT = 256>
__device__ __forceinline__ void tile_load(TileRegs<CP, NT>& t, const float* __restrict__ xb, int64_t ldx,
                                          int N, int row0, int tid = (int)threadIdx.x) {
  // The loads carry no guards and their values no select on r < N: with `ok ? loaded : 0` the compiler sinks the load
  // into a branch on ok and waits for it on the spot (s_waitcnt vmcnt(0) right behind the global_load) -- the prefetch
  // of tile t+3 then cost its full latency in every iteration of knn_pc_kernel (10-13 us per launch until round 3).
  if constexpr (CP == 4) {
    const int r = min(row0 + (tid & (TJ - 1)), N - 1);
    const float* p = xb + (int64_t)r * ldx;
    t.lo[0] = make_float4(p[0], p[1], p[2], 0.f);
  } else {
    constexpr int CH = CP / 8;                 // chunks per row
#pragma unroll
    for (int u = 0; u < TileRegs<CP, NT>::NV; ++u) {
      const int item = tid + u * NT;
      const int r = min(row0 + item / CH, N - 1), c8 = item % CH;
      const float4* p = reinterpret_cast<const float4*>(xb + (int64_t)r * ldx + c8 * 8);
      t.lo[u] = p[0];
      t.hi[u] = p[1];
    }
  }
}

// Register -> LDS half: de-interleave; with NORM also the row norms |x_j|^2 (rows >= N: +inf).
template <int CP, bool NORM = true, int NT = 256>
__device__ __forceinline__ void tile_store(const TileRegs<CP, NT>& t, float* __restrict__ s_tile,
                                           float* __restrict__ s_norm, int N, int row0, int tid = (int)threadIdx.x) {
  constexpr int RS = CP + 4;
  if constexpr (CP == 4) {
    if (tid < TJ) {
      const int r = tid;
      const float4 p = t.lo[0];
      float* d = s_tile + r * RS;
      d[0] = p.x; d[1] = p.z;        // even features 0,2
      d[2] = p.y; d[3] = 0.f;        // odd features 1,(3 = pad)
      if constexpr (NORM) s_norm[r] = (row0 + r < N) ? sq3(p.x, p.y, p.z) : INFINITY;
    }
  } else {
    constexpr int CH = CP / 8;
    constexpr int HALF = CP / 2;
#pragma unroll
    for (int u = 0; u < TileRegs<CP, NT>::NV; ++u) {
      const int item = tid + u * NT;
      const int r = item / CH, c8 = item % CH;
      const float4 a = t.lo[u], b = t.hi[u];
      float* d = s_tile + r * RS;
      *reinterpret_cast<float4*>(d + 4 * c8) = make_float4(a.x, a.z, b.x, b.z);
      *reinterpret_cast<float4*>(d + HALF + 4 * c8) = make_float4(a.y, a.w, b.y, b.w);
      if constexpr (NORM) {
        float p = __fmul_rn(a.x, a.x);
        p = fmaf(a.y, a.y, p); p = fmaf(a.z, a.z, p); p = fmaf(a.w, a.w, p);
        p = fmaf(b.x, b.x, p); p = fmaf(b.y, b.y, p); p = fmaf(b.z, b.z, p); p = fmaf(b.w, b.w, p);
        // fixed-order tree over the CH lanes of this row (consecutive lanes, CH <= 16: inside one DPP row).  The DPP
        // partners (quad_perm, row_half_mirror, row_mirror) hold the same partial sums as the xor partners of a
        // shuffle tree, so the result is bit for bit that tree's -- without four trips through the LDS crossbar.
        static_assert(CH <= 16, "one DPP row");
        if (CH > 1) p += __int_as_float(sug_dpp<0xB1, 0xf>(__float_as_int(p)));      // xor 1
        if (CH > 2) p += __int_as_float(sug_dpp<0x4E, 0xf>(__float_as_int(p)));      // xor 2
        if (CH > 4) p += __int_as_float(sug_dpp<0x141, 0xf>(__float_as_int(p)));     // other quad of the 8
        if (CH > 8) p += __int_as_float(sug_dpp<0x140, 0xf>(__float_as_int(p)));     // other half of the 16
        if (c8 == 0) s_norm[r] = (row0 + r < N) ? p : INFINITY;
      }
    }
  }
}

}  // namespace sug_tile
