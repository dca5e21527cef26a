// Farthest point sampling, ball query, kNN-of-few-queries and 3-NN.
// Reference semantics: model/point_utils.py:5-26, :86-109, :134-165 and
// model/pointnet2_utils.py:60-104 (the torch path, not the dead CUDA extension).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// FPS: one workgroup per cloud; every thread keeps PPT points and their running
// min-distance in registers, the cloud also sits in LDS so the winner's coordinates
// are a broadcast read.  One barrier per round (double-buffered wave slots).
// arg-max rule = torch.max: greatest value, lowest index among equals.
// ---------------------------------------------------------------------------
template <int PPT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float* __restrict__ xyz,
                                                    const int32_t* __restrict__ start, int N,
                                                    int npoint, int32_t* __restrict__ out) {
  extern __shared__ float s_xyz[];  // N*3
  constexpr int NW = BLOCK / WAVE;
  __shared__ float s_v[2][NW];
  __shared__ int s_i[2][NW];
  const int b = blockIdx.x;
  const float* pb = xyz + (int64_t)b * N * 3;
  for (int e = threadIdx.x; e < N * 3; e += BLOCK) s_xyz[e] = pb[e];
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int j = p * BLOCK + threadIdx.x;
    const bool ok = j < N;
    px[p] = ok ? pb[j * 3 + 0] : 0.f;
    py[p] = ok ? pb[j * 3 + 1] : 0.f;
    pz[p] = ok ? pb[j * 3 + 2] : 0.f;
    md[p] = ok ? 1e10f : -INFINITY;
  }
  int cur = start[b];
  cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  for (int i = 0; i < npoint; ++i) {
    if (threadIdx.x == 0) out[(int64_t)b * npoint + i] = cur;
    if (i == npoint - 1) break;
    const float cx = s_xyz[cur * 3 + 0], cy = s_xyz[cur * 3 + 1], cz = s_xyz[cur * 3 + 2];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float dx = __fsub_rn(px[p], cx), dy = __fsub_rn(py[p], cy), dz = __fsub_rn(pz[p], cz);
      const float d = sq3(dx, dy, dz);
      const float m = (d < md[p]) ? d : md[p];
      md[p] = m;
      if (m > bv) {  // ascending index within the thread: strict > keeps the lowest
        bv = m;
        bi = p * BLOCK + threadIdx.x;
      }
    }
    // wave arg-max: greatest value, lowest index among equals (two DPP reductions: the value, then the index)
    const float wm = wave_max_f(bv);
    bi = wave_min_i(bv == wm ? bi : 0x7fffffff);
    bv = wm;
    const int par = i & 1;
    if (lane == 0) {
      s_v[par][wv] = bv;
      s_i[par][wv] = bi;
    }
    __syncthreads();
    bv = s_v[par][0];
    bi = s_i[par][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float ov = s_v[par][w];
      const int oi = s_i[par][w];
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    cur = bi < N ? bi : 0;
  }
}

// ---------------------------------------------------------------------------
// Ball query: one wave per query, 64 candidates per step, ballot + prefix popcount
// gives the ascending-index order of the reference's sort-based implementation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz,
                                                         const float* __restrict__ qry, int N, int S,
                                                         float r2, int ns, int32_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (s >= S) return;  // whole wave exits together
  const int lane = threadIdx.x & (WAVE - 1);
  const float* pb = xyz + (int64_t)b * N * 3;
  const float* q = qry + ((int64_t)b * S + s) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  int32_t* o = out + ((int64_t)b * S + s) * ns;
  int cnt = 0, first = N;
  for (int j0 = 0; j0 < N && cnt < ns; j0 += WAVE) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < N) {
      const float x = pb[j * 3 + 0], y = pb[j * 3 + 1], z = pb[j * 3 + 2];
      const float d = sqdist_expanded(dot3(qx, qy, qz, x, y, z), nq, sq3(x, y, z));
      hit = !(d > r2);
    }
    const unsigned long long m = __ballot(hit);
    if (m) {
      if (cnt == 0) first = j0 + __builtin_ctzll(m);
      const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (hit && pos < ns) o[pos] = j;
      cnt += __builtin_popcountll(m);
    }
  }
  if (cnt > ns) cnt = ns;
  for (int p = cnt + lane; p < ns; p += WAVE) o[p] = first;
}

// ---------------------------------------------------------------------------
// k nearest candidates for few queries (full-sort semantics): one wave per query,
// the N distances live in registers (NPL per lane), k rounds of wave arg-min.
// ---------------------------------------------------------------------------
template <int NPL, bool DIRECT>
__global__ __launch_bounds__(256) void knn_query_kernel(const float* __restrict__ xyz,
                                                        const float* __restrict__ qry, int N, int S,
                                                        int k, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ dist_out) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (s >= S) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const float* pb = xyz + (int64_t)b * N * 3;
  const float* q = qry + ((int64_t)b * S + s) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  float d[NPL];
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const int j = p * WAVE + lane;
    if (j < N) {
      const float x = pb[j * 3 + 0], y = pb[j * 3 + 1], z = pb[j * 3 + 2];
      // DIRECT: sum((q - p)^2) as square_distance_Ptrans (point_utils.py:43-57 / PTran_utils.py:22-36);
      // otherwise the expanded form of square_distance (point_utils.py:112-131)
      d[p] = DIRECT ? sq3(qx - x, qy - y, qz - z) : sqdist_expanded(dot3(qx, qy, qz, x, y, z), nq, sq3(x, y, z));
    } else {
      d[p] = INFINITY;
    }
  }
  int res_i = 0;
  float res_d = 0.f;
  for (int t = 0; t < k; ++t) {
    float lv = INFINITY;
    int li = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const int j = p * WAVE + lane;
      // a taken slot is marked NaN-free +inf with index pushed out of range below
      if (d[p] < lv) {
        lv = d[p];
        li = j;
      }
    }
    // wave arg-min: smallest value, lowest index among equals (DPP reductions; all candidates +inf: index out of range)
    const float bv = wave_min_f(lv);
    const int bi = wave_min_i(lv == bv ? li : 0x7fffffff);
    if (lane == (t & (WAVE - 1))) {
      res_i = bi;
      res_d = bv;
    }
    if ((bi & (WAVE - 1)) == lane) {
      const int slot = bi / WAVE;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        if (p == slot) d[p] = INFINITY;
    }
  }
  if (lane < k) {
    idx_out[((int64_t)b * S + s) * k + lane] = res_i;
    if (dist_out) dist_out[((int64_t)b * S + s) * k + lane] = res_d;
  }
}

// ---------------------------------------------------------------------------
// 3 nearest of S candidates (LDS-resident) for every point of the dense cloud.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ qry,
                                                       const float* __restrict__ cand, int N, int S,
                                                       int32_t* __restrict__ idx3,
                                                       float* __restrict__ dist3) {
  extern __shared__ float s_c[];  // S*4: x,y,z,|c|^2
  const int b = blockIdx.y;
  const float* cb = cand + (int64_t)b * S * 3;
  for (int j = threadIdx.x; j < S; j += 256) {
    const float x = cb[j * 3 + 0], y = cb[j * 3 + 1], z = cb[j * 3 + 2];
    s_c[j * 4 + 0] = x;
    s_c[j * 4 + 1] = y;
    s_c[j * 4 + 2] = z;
    s_c[j * 4 + 3] = sq3(x, y, z);
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* q = qry + ((int64_t)b * N + n) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int j = 0; j < S; ++j) {
    const float4 c = reinterpret_cast<const float4*>(s_c)[j];
    const float d = sqdist_expanded(dot3(qx, qy, qz, c.x, c.y, c.z), nq, c.w);
    if (d < d2) {
      if (d < d1) {
        d2 = d1; i2 = i1;
        if (d < d0) {
          d1 = d0; i1 = i0;
          d0 = d; i0 = j;
        } else {
          d1 = d; i1 = j;
        }
      } else {
        d2 = d; i2 = j;
      }
    }
  }
  const int64_t o = ((int64_t)b * N + n) * 3;
  idx3[o + 0] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
  dist3[o + 0] = d0; dist3[o + 1] = d1; dist3[o + 2] = d2;
}

template <int PPT>
int launch_fps(const float* xyz, const int32_t* start, int B, int N, int npoint, int32_t* out,
               hipStream_t st) {
  constexpr int BLOCK = 256;
  size_t sh = (size_t)N * 3 * sizeof(float);
  hipLaunchKernelGGL((fps_kernel<PPT, BLOCK>), dim3(B), dim3(BLOCK), sh, st, xyz, start, N, npoint, out);
  SUG_LAUNCH_CHECK("sug_fps");
  return SUG_OK;
}

}  // namespace

extern "C" int sug_fps(const float* xyz, const int32_t* start, int B, int N, int npoint,
                       int32_t* out, void* stream) {
  SUG_REQUIRE(xyz && start && out, "sug_fps: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && npoint > 0, "sug_fps: bad shape B=%d N=%d npoint=%d", B, N, npoint);
  SUG_REQUIRE(N <= 8192, "sug_fps: N=%d > 8192", N);
  hipStream_t st = (hipStream_t)stream;
  const int ppt = sug_divup(N, 256);
  if (ppt <= 1) return launch_fps<1>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 2) return launch_fps<2>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 4) return launch_fps<4>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 8) return launch_fps<8>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 16) return launch_fps<16>(xyz, start, B, N, npoint, out, st);
  return launch_fps<32>(xyz, start, B, N, npoint, out, st);
}

extern "C" int sug_ball_query(const float* xyz, const float* query, int B, int N, int S, float r2,
                              int nsample, int32_t* out, void* stream) {
  SUG_REQUIRE(xyz && query && out, "sug_ball_query: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && nsample > 0, "sug_ball_query: bad shape");
  SUG_REQUIRE(B <= 65535, "sug_ball_query: B too large");
  dim3 grid(sug_divup(S, 256 / WAVE), B);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, query, N, S, r2,
                     nsample, out);
  SUG_LAUNCH_CHECK("sug_ball_query");
  return SUG_OK;
}

template <bool DIRECT>
static int launch_knn_query(const float* xyz, const float* query, int B, int N, int S, int k, int32_t* idx_out,
                            float* dist_out, void* stream) {
  SUG_REQUIRE(xyz && query && idx_out, "sug_knn_query: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0, "sug_knn_query: bad shape");
  SUG_REQUIRE(k >= 1 && k <= 64 && k <= N, "sug_knn_query: need 1 <= k <= min(64,N), got %d", k);
  SUG_REQUIRE(N <= 4096, "sug_knn_query: N=%d > 4096", N);
  SUG_REQUIRE(B <= 65535, "sug_knn_query: B too large");
  dim3 grid(sug_divup(S, 256 / WAVE), B);
  hipStream_t st = (hipStream_t)stream;
  const int npl = sug_divup(N, WAVE);
  if (npl <= 4)
    hipLaunchKernelGGL((knn_query_kernel<4, DIRECT>), grid, dim3(256), 0, st, xyz, query, N, S, k, idx_out, dist_out);
  else if (npl <= 8)
    hipLaunchKernelGGL((knn_query_kernel<8, DIRECT>), grid, dim3(256), 0, st, xyz, query, N, S, k, idx_out, dist_out);
  else if (npl <= 16)
    hipLaunchKernelGGL((knn_query_kernel<16, DIRECT>), grid, dim3(256), 0, st, xyz, query, N, S, k, idx_out, dist_out);
  else if (npl <= 32)
    hipLaunchKernelGGL((knn_query_kernel<32, DIRECT>), grid, dim3(256), 0, st, xyz, query, N, S, k, idx_out, dist_out);
  else
    hipLaunchKernelGGL((knn_query_kernel<64, DIRECT>), grid, dim3(256), 0, st, xyz, query, N, S, k, idx_out, dist_out);
  SUG_LAUNCH_CHECK("sug_knn_query");
  return SUG_OK;
}

extern "C" int sug_knn_query(const float* xyz, const float* query, int B, int N, int S, int k,
                             int32_t* idx_out, float* dist_out, void* stream) {
  return launch_knn_query<false>(xyz, query, B, N, S, k, idx_out, dist_out, stream);
}

extern "C" int sug_knn_query_direct(const float* xyz, const float* query, int B, int N, int S, int k,
                                    int32_t* idx_out, float* dist_out, void* stream) {
  return launch_knn_query<true>(xyz, query, B, N, S, k, idx_out, dist_out, stream);
}

extern "C" int sug_three_nn(const float* query, const float* cand, int B, int N, int S,
                            int32_t* idx3, float* dist3, void* stream) {
  SUG_REQUIRE(query && cand && idx3 && dist3, "sug_three_nn: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S >= 3, "sug_three_nn: bad shape (S=%d must be >= 3)", S);
  SUG_REQUIRE(S <= 2048, "sug_three_nn: S=%d > 2048", S);
  SUG_REQUIRE(B <= 65535, "sug_three_nn: B too large");
  dim3 grid(sug_divup(N, 256), B);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), (size_t)S * 4 * sizeof(float),
                     (hipStream_t)stream, query, cand, N, S, idx3, dist3);
  SUG_LAUNCH_CHECK("sug_three_nn");
  return SUG_OK;
}
