// SA-node module glue (adapt_layer_off, model/model_utils.py:103-128) as two fused kernels
// with hand-written backward passes; replaces ~35 forward + ~50 backward elementwise/gather
// launches per encoder pass.
//   node_offset : off[b,s,:] = mean_j tanh(proj[g_j] - proj[f]) * (loc[g_j] - loc[f]),
//                 nloc = loc[f] + off            (model_utils.py:110-119; proj = fea . W_off^T)
//   interp3_cat : out[b,n,:] = [ fea[b,n,:] | sum_t w_t * node[b, idx3[n,t], :] ],
//                 w_t = (1/max(d_t,1e-10)) / sum   (upsample_inter, model/point_utils.py:156-162)
// All tensors are tiny (B*64 nodes, B*N points x 64 channels): latency-bound, one launch each.
#include "common.h"

namespace {

// One wave per node (b,s); lanes run over the ns neighbours, wave shuffle reduction.
__global__ __launch_bounds__(256) void node_offset_fwd_kernel(const float* __restrict__ proj,
                                                              const float* __restrict__ loc,
                                                              const int32_t* __restrict__ fidx,
                                                              const int32_t* __restrict__ gidx, int N,
                                                              int S, int ns, int total,
                                                              float* __restrict__ off,
                                                              float* __restrict__ nloc) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, s)
  if (e >= total) return;
  const int lane = threadIdx.x & 63;
  const int b = e / S;
  const float* pb = proj + (int64_t)b * N * 3;
  const float* lb = loc + (int64_t)b * N * 3;
  int f = fidx[e];
  f = f < 0 ? 0 : (f >= N ? N - 1 : f);
  const float pcx = pb[f * 3 + 0], pcy = pb[f * 3 + 1], pcz = pb[f * 3 + 2];
  const float lcx = lb[f * 3 + 0], lcy = lb[f * 3 + 1], lcz = lb[f * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  const int32_t* g = gidx + (int64_t)e * ns;
  for (int j = lane; j < ns; j += 64) {
    const int n = g[j];
    if (n < 0 || n >= N) continue;                    // zero-hit rows of the ball query
    ax += tanhf(pb[n * 3 + 0] - pcx) * (lb[n * 3 + 0] - lcx);
    ay += tanhf(pb[n * 3 + 1] - pcy) * (lb[n * 3 + 1] - lcy);
    az += tanhf(pb[n * 3 + 2] - pcz) * (lb[n * 3 + 2] - lcz);
  }
  ax = wave_sum_f(ax); ay = wave_sum_f(ay); az = wave_sum_f(az);
  if (lane == 0) {
    const float inv = 1.0f / (float)ns;
    ax *= inv; ay *= inv; az *= inv;
    off[e * 3 + 0] = ax; off[e * 3 + 1] = ay; off[e * 3 + 2] = az;
    nloc[e * 3 + 0] = lcx + ax; nloc[e * 3 + 1] = lcy + ay; nloc[e * 3 + 2] = lcz + az;
  }
}

// dproj[g_j] += g * d_j * (1 - t^2) / ns ; dproj[f] -= sum_j (same)   (dproj zeroed by the entry point)
__global__ __launch_bounds__(256) void node_offset_bwd_kernel(const float* __restrict__ proj,
                                                              const float* __restrict__ loc,
                                                              const int32_t* __restrict__ fidx,
                                                              const int32_t* __restrict__ gidx,
                                                              const float* __restrict__ goff, int N,
                                                              int S, int ns, int total,
                                                              float* __restrict__ dproj) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= total) return;
  const int lane = threadIdx.x & 63;
  const int b = e / S;
  const float* pb = proj + (int64_t)b * N * 3;
  const float* lb = loc + (int64_t)b * N * 3;
  float* db = dproj + (int64_t)b * N * 3;
  int f = fidx[e];
  f = f < 0 ? 0 : (f >= N ? N - 1 : f);
  const float pcx = pb[f * 3 + 0], pcy = pb[f * 3 + 1], pcz = pb[f * 3 + 2];
  const float lcx = lb[f * 3 + 0], lcy = lb[f * 3 + 1], lcz = lb[f * 3 + 2];
  const float inv = 1.0f / (float)ns;
  const float gx = goff[e * 3 + 0] * inv, gy = goff[e * 3 + 1] * inv, gz = goff[e * 3 + 2] * inv;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  const int32_t* g = gidx + (int64_t)e * ns;
  for (int j = lane; j < ns; j += 64) {
    const int n = g[j];
    if (n < 0 || n >= N) continue;
    const float tx = tanhf(pb[n * 3 + 0] - pcx), ty = tanhf(pb[n * 3 + 1] - pcy), tz = tanhf(pb[n * 3 + 2] - pcz);
    const float vx = gx * (lb[n * 3 + 0] - lcx) * (1.f - tx * tx);
    const float vy = gy * (lb[n * 3 + 1] - lcy) * (1.f - ty * ty);
    const float vz = gz * (lb[n * 3 + 2] - lcz) * (1.f - tz * tz);
    atomicAdd(&db[n * 3 + 0], vx); atomicAdd(&db[n * 3 + 1], vy); atomicAdd(&db[n * 3 + 2], vz);
    cx += vx; cy += vy; cz += vz;
  }
  cx = wave_sum_f(cx); cy = wave_sum_f(cy); cz = wave_sum_f(cz);
  if (lane == 0) {
    atomicAdd(&db[f * 3 + 0], -cx); atomicAdd(&db[f * 3 + 1], -cy); atomicAdd(&db[f * 3 + 2], -cz);
  }
}

// The same per cloud with the [N,3] gradient in LDS: one workgroup per cloud, a wave per node (786 k global float
// atomics at the C2 shape ran at 82 us; LDS adds + one plain write of the cloud's dproj: the result needs no zero fill)
__global__ __launch_bounds__(1024) void node_offset_bwd_lds_kernel(const float* __restrict__ proj,
                                                                   const float* __restrict__ loc,
                                                                   const int32_t* __restrict__ fidx,
                                                                   const int32_t* __restrict__ gidx,
                                                                   const float* __restrict__ goff, int N, int S,
                                                                   int ns, float* __restrict__ dproj) {
  extern __shared__ float s_d[];                 // [N*3]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < N * 3; i += 1024) s_d[i] = 0.f;
  __syncthreads();
  const float* pb = proj + (int64_t)b * N * 3;
  const float* lb = loc + (int64_t)b * N * 3;
  const float inv = 1.0f / (float)ns;
  for (int sn = wave; sn < S; sn += 16) {
    const int e = b * S + sn;
    int f = fidx[e];
    f = f < 0 ? 0 : (f >= N ? N - 1 : f);
    const float pcx = pb[f * 3 + 0], pcy = pb[f * 3 + 1], pcz = pb[f * 3 + 2];
    const float lcx = lb[f * 3 + 0], lcy = lb[f * 3 + 1], lcz = lb[f * 3 + 2];
    const float gx = goff[e * 3 + 0] * inv, gy = goff[e * 3 + 1] * inv, gz = goff[e * 3 + 2] * inv;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    const int32_t* g = gidx + (int64_t)e * ns;
    for (int j = lane; j < ns; j += 64) {
      const int n = g[j];
      if (n < 0 || n >= N) continue;
      const float tx = tanhf(pb[n * 3 + 0] - pcx), ty = tanhf(pb[n * 3 + 1] - pcy), tz = tanhf(pb[n * 3 + 2] - pcz);
      const float vx = gx * (lb[n * 3 + 0] - lcx) * (1.f - tx * tx);
      const float vy = gy * (lb[n * 3 + 1] - lcy) * (1.f - ty * ty);
      const float vz = gz * (lb[n * 3 + 2] - lcz) * (1.f - tz * tz);
      atomicAdd(&s_d[n * 3 + 0], vx); atomicAdd(&s_d[n * 3 + 1], vy); atomicAdd(&s_d[n * 3 + 2], vz);
      cx += vx; cy += vy; cz += vz;
    }
    cx = wave_sum_f(cx); cy = wave_sum_f(cy); cz = wave_sum_f(cz);
    if (lane == 0) {
      atomicAdd(&s_d[f * 3 + 0], -cx); atomicAdd(&s_d[f * 3 + 1], -cy); atomicAdd(&s_d[f * 3 + 2], -cz);
    }
  }
  __syncthreads();
  float* db = dproj + (int64_t)b * N * 3;
  for (int i = threadIdx.x; i < N * 3; i += 1024) db[i] = s_d[i];
}

// The same without float atomics (the default whenever P >= 1 accumulator planes of the cloud's [N,3] gradient fit LDS): wave w
// of the workgroup owns the nodes w, w + P, ... and its own plane, and takes them one after the other; inside a node the
// lanes are its neighbours (64 per trip).  Lanes that name the same point (the ball query pads a short list by repeating
// its first hit) are combined at the LOWEST such lane in ascending lane order -- found with an LDS integer atomicMin on a
// per-wave tag table, the duplicates then walked with uniform readlanes -- and that lane adds to the plane with a plain
// read-modify-write; the node's centre term -sum_j v (lane-strided sums, fixed xor tree) follows.  At the end the planes
// are folded in plane order.  Every sum has one fixed order: the gradient is reproducible bit for bit (the LDS-atomic
// kernel above left the order to the hardware, as the reference's index_points backward does).
template <int P>
__global__ __launch_bounds__(64 * P) void node_offset_bwd_ordered_kernel(const float* __restrict__ proj,
                                                                         const float* __restrict__ loc,
                                                                         const int32_t* __restrict__ fidx,
                                                                         const int32_t* __restrict__ gidx,
                                                                         const float* __restrict__ goff, int N, int S,
                                                                         int ns, float* __restrict__ dproj) {
  extern __shared__ __attribute__((aligned(16))) float s_o[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* plane = s_o + (size_t)wave * 3 * N;                                   // [P][N*3]
  int* tag = reinterpret_cast<int*>(s_o + (size_t)P * 3 * N) + (size_t)wave * N;   // [P][N]
  const int b = blockIdx.x;
  for (int i = lane; i < 3 * N; i += 64) plane[i] = 0.f;
  __builtin_amdgcn_wave_barrier();
  const float* pb = proj + (int64_t)b * N * 3;
  const float* lb = loc + (int64_t)b * N * 3;
  const float inv = 1.0f / (float)ns;
  // one node's lanes: dedupe (lowest lane of a point collects its duplicates in lane order), add to the plane
  auto add_node = [&](int n, float vx, float vy, float vz) {
    if (n >= 0) tag[n] = 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    if (n >= 0) atomicMin(&tag[n], lane);
    __builtin_amdgcn_wave_barrier();
    const int leader = n >= 0 ? tag[n] : lane;
    unsigned long long dup = __ballot(leader != lane);
    float ax = vx, ay = vy, az = vz;
    while (dup) {
      const int i = __builtin_ctzll(dup);
      dup &= dup - 1;
      const int li = __builtin_amdgcn_readlane(leader, i);
      const float dx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx), i));
      const float dy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy), i));
      const float dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz), i));
      const bool mine = lane == li;
      ax = mine ? ax + dx : ax; ay = mine ? ay + dy : ay; az = mine ? az + dz : az;
    }
    if (n >= 0 && leader == lane) {
      plane[n * 3 + 0] += ax; plane[n * 3 + 1] += ay; plane[n * 3 + 2] += az;
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto centre = [&](int f, float cx, float cy, float cz) {
    cx = wave_sum_f(cx); cy = wave_sum_f(cy); cz = wave_sum_f(cz);
    if (lane == 0) {
      plane[f * 3 + 0] -= cx; plane[f * 3 + 1] -= cy; plane[f * 3 + 2] -= cz;
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (ns <= 64) {
    // one trip per node; the loads run two nodes ahead of the arithmetic: indices (centre point, this lane's neighbour,
    // the node's offset gradient) of node k+2 and the gathered rows of node k+1 are in flight while node k is added
    struct Idx { int f, n; float gx, gy, gz; };
    struct Val { float pc[3], lc[3], pn[3], ln[3]; };
    auto load_idx = [&](int sn) {
      Idx r;
      const int e = b * S + sn;
      int f = fidx[e];
      r.f = f < 0 ? 0 : (f >= N ? N - 1 : f);
      int n = lane < ns ? gidx[(int64_t)e * ns + lane] : -1;
      r.n = n >= N ? -1 : n;
      r.gx = goff[e * 3 + 0] * inv; r.gy = goff[e * 3 + 1] * inv; r.gz = goff[e * 3 + 2] * inv;
      return r;
    };
    auto load_val = [&](const Idx& ix) {
      Val v;
      const int n = ix.n >= 0 ? ix.n : ix.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v.pc[d] = pb[ix.f * 3 + d]; v.lc[d] = lb[ix.f * 3 + d];
        v.pn[d] = pb[n * 3 + d];    v.ln[d] = lb[n * 3 + d];
      }
      return v;
    };
    Idx i0 = {}, i1 = {};
    Val v0 = {};
    if (wave < S) { i0 = load_idx(wave); v0 = load_val(i0); }
    if (wave + P < S) i1 = load_idx(wave + P);
    for (int sn = wave; sn < S; sn += P) {
      const Idx ci = i0;
      const Val cv = v0;
      i0 = i1;
      if (sn + P < S) v0 = load_val(i0);
      if (sn + 2 * P < S) i1 = load_idx(sn + 2 * P);
      float vx = 0.f, vy = 0.f, vz = 0.f;
      if (ci.n >= 0) {
        const float tx = tanhf(cv.pn[0] - cv.pc[0]), ty = tanhf(cv.pn[1] - cv.pc[1]), tz = tanhf(cv.pn[2] - cv.pc[2]);
        vx = ci.gx * (cv.ln[0] - cv.lc[0]) * (1.f - tx * tx);
        vy = ci.gy * (cv.ln[1] - cv.lc[1]) * (1.f - ty * ty);
        vz = ci.gz * (cv.ln[2] - cv.lc[2]) * (1.f - tz * tz);
      }
      add_node(ci.n, vx, vy, vz);
      centre(ci.f, vx, vy, vz);
    }
  } else {
    for (int sn = wave; sn < S; sn += P) {
      const int e = b * S + sn;
      int f = fidx[e];
      f = f < 0 ? 0 : (f >= N ? N - 1 : f);
      const float pcx = pb[f * 3 + 0], pcy = pb[f * 3 + 1], pcz = pb[f * 3 + 2];
      const float lcx = lb[f * 3 + 0], lcy = lb[f * 3 + 1], lcz = lb[f * 3 + 2];
      const float gx = goff[e * 3 + 0] * inv, gy = goff[e * 3 + 1] * inv, gz = goff[e * 3 + 2] * inv;
      float cx = 0.f, cy = 0.f, cz = 0.f;
      const int32_t* g = gidx + (int64_t)e * ns;
      for (int j0 = 0; j0 < ns; j0 += 64) {
        const int j = j0 + lane;
        int n = j < ns ? g[j] : -1;
        if (n >= N) n = -1;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        if (n >= 0) {
          const float tx = tanhf(pb[n * 3 + 0] - pcx), ty = tanhf(pb[n * 3 + 1] - pcy), tz = tanhf(pb[n * 3 + 2] - pcz);
          vx = gx * (lb[n * 3 + 0] - lcx) * (1.f - tx * tx);
          vy = gy * (lb[n * 3 + 1] - lcy) * (1.f - ty * ty);
          vz = gz * (lb[n * 3 + 2] - lcz) * (1.f - tz * tz);
        }
        cx += vx; cy += vy; cz += vz;
        add_node(n, vx, vy, vz);
      }
      centre(f, cx, cy, cz);
    }
  }
  __syncthreads();
  float* db = dproj + (int64_t)b * N * 3;
  for (int i = threadIdx.x; i < N * 3; i += 64 * P) {
    float a = s_o[i];
#pragma unroll
    for (int p = 1; p < P; ++p) a += s_o[(size_t)p * 3 * N + i];
    db[i] = a;
  }
}

// 16 lanes x float4 per point (C2 = 64 interpolated channels); generic C2 % 4 == 0 via a loop.
__global__ __launch_bounds__(256) void interp3_cat_fwd_kernel(const float* __restrict__ fea, int64_t ldf,
                                                              int C1, const float* __restrict__ node,
                                                              const int32_t* __restrict__ idx3,
                                                              const float* __restrict__ d3, int N, int S,
                                                              int C2, int64_t BN, float* __restrict__ out,
                                                              int64_t ldo) {
  const int64_t p = (int64_t)blockIdx.x * 16 + threadIdx.x / 16;
  if (p >= BN) return;
  const int lane = threadIdx.x & 15;
  const int64_t b = p / N;
  const float r0 = 1.0f / fmaxf(d3[p * 3 + 0], 1e-10f), r1 = 1.0f / fmaxf(d3[p * 3 + 1], 1e-10f),
              r2 = 1.0f / fmaxf(d3[p * 3 + 2], 1e-10f);
  const float R = (r0 + r1) + r2;
  const float w0 = r0 / R, w1 = r1 / R, w2 = r2 / R;
  const float* n0 = node + (b * S + idx3[p * 3 + 0]) * C2;
  const float* n1 = node + (b * S + idx3[p * 3 + 1]) * C2;
  const float* n2 = node + (b * S + idx3[p * 3 + 2]) * C2;
  for (int c = lane * 4; c < C1; c += 64)
    *reinterpret_cast<float4*>(out + p * ldo + c) = *reinterpret_cast<const float4*>(fea + p * ldf + c);
  for (int c = lane * 4; c < C2; c += 64) {
    const float4 a = *reinterpret_cast<const float4*>(n0 + c), bq = *reinterpret_cast<const float4*>(n1 + c),
                 cq = *reinterpret_cast<const float4*>(n2 + c);
    float4 o;      // torch.sum(sel * w, dim=2): ((a*w0 + b*w1) + c*w2)
    o.x = (a.x * w0 + bq.x * w1) + cq.x * w2; o.y = (a.y * w0 + bq.y * w1) + cq.y * w2;
    o.z = (a.z * w0 + bq.z * w1) + cq.z * w2; o.w = (a.w * w0 + bq.w * w1) + cq.w * w2;
    *reinterpret_cast<float4*>(out + p * ldo + C1 + c) = o;
  }
}

// g: gradient of out [B,N,C1+C2] (row stride ldg).  dnode += w_t*g_interp,
// dd_t = -(r_t^2/R) * <g_interp, node_t - interp>  (0 where d_t was clamped), and through
// d_t = |q - c_t|^2:  dnloc[idx_t] += dd_t * 2*(nloc[idx_t] - xyz[n]).
// One workgroup = one cloud x one chunk of its points; the cloud's node gradients [S,C2]
// (16 KB at 64x64) and node-position gradients accumulate in LDS (every node receives ~48
// contributions per channel: global atomics would serialise) and are flushed once.
__global__ __launch_bounds__(256) void interp3_cat_bwd_kernel(const float* __restrict__ g, int64_t ldg,
                                                              int C1, const float* __restrict__ node,
                                                              const int32_t* __restrict__ idx3,
                                                              const float* __restrict__ d3,
                                                              const float* __restrict__ xyz,
                                                              const float* __restrict__ nloc, int N,
                                                              int S, int C2, int chunks,
                                                              float* __restrict__ dnode,
                                                              float* __restrict__ dnloc) {
  extern __shared__ float s_acc[];               // [S*C2] node grads | [S*3] node-position grads
  float* s_dn = s_acc;
  float* s_dl = s_acc + S * C2;
  const int b = blockIdx.y, ck = blockIdx.x;
  for (int i = threadIdx.x; i < S * C2 + S * 3; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int n0 = (int)((int64_t)N * ck / chunks), n1 = (int)((int64_t)N * (ck + 1) / chunks);
  const float* nodeb = node + (int64_t)b * S * C2;
  for (int n = n0 + grp; n < n1 + 15; n += 16) {   // all 16 lanes of a group share n (uniform trip count)
    const bool live = n < n1;
    const int64_t p = (int64_t)b * N + (live ? n : n0);
    const float d0 = d3[p * 3 + 0], d1 = d3[p * 3 + 1], d2 = d3[p * 3 + 2];
    const float r0 = 1.0f / fmaxf(d0, 1e-10f), r1 = 1.0f / fmaxf(d1, 1e-10f), r2 = 1.0f / fmaxf(d2, 1e-10f);
    const float R = (r0 + r1) + r2;
    const float w0 = r0 / R, w1 = r1 / R, w2 = r2 / R;
    const int i0 = idx3[p * 3 + 0], i1 = idx3[p * 3 + 1], i2 = idx3[p * 3 + 2];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (live) {
      for (int c = lane * 4; c < C2; c += 64) {
        const float4 gv = *reinterpret_cast<const float4*>(g + p * ldg + C1 + c);
        const float4 a = *reinterpret_cast<const float4*>(nodeb + i0 * C2 + c);
        const float4 bq = *reinterpret_cast<const float4*>(nodeb + i1 * C2 + c);
        const float4 cq = *reinterpret_cast<const float4*>(nodeb + i2 * C2 + c);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
        const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {bq.x, bq.y, bq.z, bq.w}, cc[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float o = (aa[v] * w0 + bb[v] * w1) + cc[v] * w2;
          s0 += gg[v] * (aa[v] - o); s1 += gg[v] * (bb[v] - o); s2 += gg[v] * (cc[v] - o);
          atomicAdd(&s_dn[i0 * C2 + c + v], w0 * gg[v]);
          atomicAdd(&s_dn[i1 * C2 + c + v], w1 * gg[v]);
          atomicAdd(&s_dn[i2 * C2 + c + v], w2 * gg[v]);
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
    }
    if (live && lane < 3) {
      const float sv = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
      const float d = lane == 0 ? d0 : (lane == 1 ? d1 : d2);
      const float r = lane == 0 ? r0 : (lane == 1 ? r1 : r2);
      const int i = lane == 0 ? i0 : (lane == 1 ? i1 : i2);
      const float dd = (d < 1e-10f) ? 0.f : -(r * r / R) * sv;     // clamp region: no gradient
      const float* q = xyz + p * 3;
      const float* cpos = nloc + ((int64_t)b * S + i) * 3;
      atomicAdd(&s_dl[i * 3 + 0], dd * 2.f * (cpos[0] - q[0]));
      atomicAdd(&s_dl[i * 3 + 1], dd * 2.f * (cpos[1] - q[1]));
      atomicAdd(&s_dl[i * 3 + 2], dd * 2.f * (cpos[2] - q[2]));
    }
  }
  __syncthreads();
  float* dnb = dnode + (int64_t)b * S * C2;
  float* dlb = dnloc + (int64_t)b * S * 3;
  for (int i = threadIdx.x; i < S * C2; i += 256) atomicAdd(&dnb[i], s_dn[i]);
  for (int i = threadIdx.x; i < S * 3; i += 256) atomicAdd(&dlb[i], s_dl[i]);
}

// interp3_cat backward without atomics, in two launches over the reverse lists of idx3 (which points interpolate from
// which node: sug_reverse_lists, sorted, so the sums have a fixed order):
//   A (per point):  o = sum_t w_t node_t,  s_t = sum_c g_c (node_t,c - o_c),  dd_t = -(r_t^2 / R) s_t  -> ddw[p] = dd | w
//   B (per node s): dnode[s,:] = sum_{(n,t) -> s} w_t(n) g[n, C1:],  dnloc[s] = sum dd_t(n) * 2 (nloc[s] - xyz[n])
// (the LDS-accumulating kernel above spends its time in 12.6 M LDS float adds: 87 us at the C2 shape)
__global__ __launch_bounds__(256) void interp3_bwd_point_kernel(const float* __restrict__ g, int64_t ldg, int C1,
                                                                const float* __restrict__ node,
                                                                const int32_t* __restrict__ idx3,
                                                                const float* __restrict__ d3, int N, int S, int C2,
                                                                int64_t BN, float* __restrict__ ddw) {
  const int64_t p = (int64_t)blockIdx.x * 16 + threadIdx.x / 16;
  if (p >= BN) return;
  const int lane = threadIdx.x & 15;
  const int64_t b = p / N;
  const float d0 = d3[p * 3 + 0], d1 = d3[p * 3 + 1], d2 = d3[p * 3 + 2];
  const float r0 = 1.0f / fmaxf(d0, 1e-10f), r1 = 1.0f / fmaxf(d1, 1e-10f), r2 = 1.0f / fmaxf(d2, 1e-10f);
  const float R = (r0 + r1) + r2;
  const float w0 = r0 / R, w1 = r1 / R, w2 = r2 / R;
  const float* n0 = node + (b * S + idx3[p * 3 + 0]) * C2;
  const float* n1 = node + (b * S + idx3[p * 3 + 1]) * C2;
  const float* n2 = node + (b * S + idx3[p * 3 + 2]) * C2;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < C2; c += 64) {
    const float4 gv = *reinterpret_cast<const float4*>(g + p * ldg + C1 + c);
    const float4 a = *reinterpret_cast<const float4*>(n0 + c), bq = *reinterpret_cast<const float4*>(n1 + c),
                 cq = *reinterpret_cast<const float4*>(n2 + c);
    const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
    const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {bq.x, bq.y, bq.z, bq.w}, cc[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float o = (aa[v] * w0 + bb[v] * w1) + cc[v] * w2;
      s0 += gg[v] * (aa[v] - o); s1 += gg[v] * (bb[v] - o); s2 += gg[v] * (cc[v] - o);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
  }
  if (lane < 3) {
    const float sv = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
    const float d = lane == 0 ? d0 : (lane == 1 ? d1 : d2);
    const float r = lane == 0 ? r0 : (lane == 1 ? r1 : r2);
    const float w = lane == 0 ? w0 : (lane == 1 ? w1 : w2);
    ddw[p * 6 + lane] = (d < 1e-10f) ? 0.f : -(r * r / R) * sv;     // clamp region: no gradient
    ddw[p * 6 + 3 + lane] = w;
  }
}

__global__ __launch_bounds__(256) void interp3_bwd_node_kernel(const float* __restrict__ g, int64_t ldg, int C1,
                                                               const float* __restrict__ ddw,
                                                               const int32_t* __restrict__ rev_off,
                                                               const int32_t* __restrict__ rev_ent,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ nloc, int N, int S, int C2,
                                                               int64_t BS, float* __restrict__ dnode,
                                                               float* __restrict__ dnloc) {
  const int64_t q = (int64_t)blockIdx.x * 16 + threadIdx.x / 16;    // node (b, s)
  if (q >= BS) return;
  const int lane = threadIdx.x & 15;
  const int64_t b = q / S;
  const int sn = (int)(q - b * S);
  const int32_t* offb = rev_off + b * (S + 1);
  const int off = offb[sn], cnt = offb[sn + 1] - off;
  const int32_t* ent = rev_ent + b * 3 * N + off;
  const int ax = lane < 3 ? lane : 0;
  const float cpos = nloc[q * 3 + ax];
  float dl = 0.f;
  const int npass = (C2 + 63) / 64;                                 // one pass for C2 <= 64 (the SA-node shapes)
  for (int pass = 0; pass < npass; ++pass) {
    const int c0 = pass * 64 + lane * 4;
    const bool chan = c0 < C2;
    float4 acc = make_float4(0, 0, 0, 0);
    constexpr int JB = 8;
    for (int t0 = 0; t0 < cnt; t0 += JB) {
      int en[JB];
      float4 gv[JB];
      float wv[JB], dv[JB], xv[JB];
#pragma unroll
      for (int t = 0; t < JB; ++t) en[t] = ent[t0 + t < cnt ? t0 + t : cnt - 1];
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        const int n = en[t] / 3, tt = en[t] - 3 * n;
        const int64_t p = b * N + n;
        gv[t] = chan ? *reinterpret_cast<const float4*>(g + p * ldg + C1 + c0) : make_float4(0, 0, 0, 0);
        wv[t] = ddw[p * 6 + 3 + tt];
        dv[t] = ddw[p * 6 + tt];
        xv[t] = xyz[p * 3 + ax];
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (t0 + t >= cnt) continue;
        acc.x = fmaf(wv[t], gv[t].x, acc.x); acc.y = fmaf(wv[t], gv[t].y, acc.y);
        acc.z = fmaf(wv[t], gv[t].z, acc.z); acc.w = fmaf(wv[t], gv[t].w, acc.w);
        if (pass == 0) dl += dv[t] * 2.f * (cpos - xv[t]);
      }
    }
    if (chan) *reinterpret_cast<float4*>(dnode + q * C2 + c0) = acc;
  }
  if (lane < 3) dnloc[q * 3 + lane] = dl;
}

}  // namespace

extern "C" int sug_node_offset_fwd(const float* proj, const float* loc, const int32_t* fidx,
                                   const int32_t* gidx, int B, int N, int S, int ns, float* off,
                                   float* nloc, void* stream) {
  SUG_REQUIRE(proj && loc && fidx && gidx && off && nloc, "sug_node_offset_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0, "sug_node_offset_fwd: bad shape");
  const int total = B * S;
  hipLaunchKernelGGL(node_offset_fwd_kernel, dim3(sug_divup(total, 4)), dim3(256), 0, (hipStream_t)stream,
                     proj, loc, fidx, gidx, N, S, ns, total, off, nloc);
  SUG_LAUNCH_CHECK("sug_node_offset_fwd");
  return SUG_OK;
}

extern "C" int sug_node_offset_bwd(const float* proj, const float* loc, const int32_t* fidx,
                                   const int32_t* gidx, const float* goff, int B, int N, int S, int ns,
                                   float* dproj, void* stream) {
  SUG_REQUIRE(proj && loc && fidx && gidx && goff && dproj, "sug_node_offset_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0, "sug_node_offset_bwd: bad shape");
  const int total = B * S;
  hipStream_t st = (hipStream_t)stream;
  {                                                          // accumulator planes of a cloud fit LDS: fixed summation order
    static const int unordered = getenv("SUG_NODE_OFFSET_UNORDERED") ? atoi(getenv("SUG_NODE_OFFSET_UNORDERED")) : 0;
    const size_t per_plane = (size_t)N * 4 * sizeof(float);        // [N,3] floats + [N] tags
    // (more than 64 KB of dynamic LDS needs the opt-in; on a device that refuses it -- none of the gfx950 parts -- the call
    // fails with the driver's message rather than silently changing the summation order)
    if (!unordered && per_plane <= 150 * 1024) {
      if (8 * per_plane <= 150 * 1024) {
        static SugLdsOptIn note;
        if (int rc = sug_allow_dynamic_lds(note, &node_offset_bwd_ordered_kernel<8>, 150 * 1024, "sug_node_offset_bwd")) return rc;
        hipLaunchKernelGGL(node_offset_bwd_ordered_kernel<8>, dim3(B), dim3(512), 8 * per_plane, st, proj, loc, fidx, gidx, goff, N,
                           S, ns, dproj);
      } else if (4 * per_plane <= 150 * 1024) {
        static SugLdsOptIn note;
        if (int rc = sug_allow_dynamic_lds(note, &node_offset_bwd_ordered_kernel<4>, 150 * 1024, "sug_node_offset_bwd")) return rc;
        hipLaunchKernelGGL(node_offset_bwd_ordered_kernel<4>, dim3(B), dim3(256), 4 * per_plane, st, proj, loc, fidx, gidx, goff, N,
                           S, ns, dproj);
      } else if (2 * per_plane <= 150 * 1024) {
        static SugLdsOptIn note;
        if (int rc = sug_allow_dynamic_lds(note, &node_offset_bwd_ordered_kernel<2>, 150 * 1024, "sug_node_offset_bwd")) return rc;
        hipLaunchKernelGGL(node_offset_bwd_ordered_kernel<2>, dim3(B), dim3(128), 2 * per_plane, st, proj, loc, fidx, gidx, goff, N,
                           S, ns, dproj);
      } else {
        static SugLdsOptIn note;
        if (int rc = sug_allow_dynamic_lds(note, &node_offset_bwd_ordered_kernel<1>, 150 * 1024, "sug_node_offset_bwd")) return rc;
        hipLaunchKernelGGL(node_offset_bwd_ordered_kernel<1>, dim3(B), dim3(64), per_plane, st, proj, loc, fidx, gidx, goff, N, S,
                           ns, dproj);
      }
      SUG_LAUNCH_CHECK("sug_node_offset_bwd");
      return SUG_OK;
    }
  }
  if ((size_t)N * 3 * sizeof(float) <= 60 * 1024) {          // the cloud's gradient fits LDS: no global atomics
    hipLaunchKernelGGL(node_offset_bwd_lds_kernel, dim3(B), dim3(1024), (size_t)N * 3 * sizeof(float), st, proj, loc, fidx,
                       gidx, goff, N, S, ns, dproj);
    SUG_LAUNCH_CHECK("sug_node_offset_bwd");
    return SUG_OK;
  }
  if (hipMemsetAsync(dproj, 0, (size_t)B * N * 3 * sizeof(float), st) != hipSuccess) {
    sug_set_error("sug_node_offset_bwd: memset failed");
    return SUG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(node_offset_bwd_kernel, dim3(sug_divup(total, 4)), dim3(256), 0, st,
                     proj, loc, fidx, gidx, goff, N, S, ns, total, dproj);
  SUG_LAUNCH_CHECK("sug_node_offset_bwd");
  return SUG_OK;
}

extern "C" int sug_interp3_cat_fwd(const float* fea, int64_t ldf, int C1, const float* node,
                                   const int32_t* idx3, const float* d3, int B, int N, int S, int C2,
                                   float* out, int64_t ldo, void* stream) {
  SUG_REQUIRE(fea && node && idx3 && d3 && out, "sug_interp3_cat_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S >= 3 && C1 >= 0 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ldf % 4 == 0 &&
                  ldo % 4 == 0 && ldo >= C1 + C2 && ldf >= C1,
              "sug_interp3_cat_fwd: bad shape (channel counts and strides must be multiples of 4)");
  SUG_REQUIRE(((uintptr_t)fea % 16 == 0) && ((uintptr_t)node % 16 == 0) && ((uintptr_t)out % 16 == 0),
              "sug_interp3_cat_fwd: pointers must be 16-byte aligned");
  const int64_t BN = (int64_t)B * N;
  hipLaunchKernelGGL(interp3_cat_fwd_kernel, dim3(sug_divup(BN, 16)), dim3(256), 0, (hipStream_t)stream, fea,
                     ldf, C1, node, idx3, d3, N, S, C2, BN, out, ldo);
  SUG_LAUNCH_CHECK("sug_interp3_cat_fwd");
  return SUG_OK;
}

extern "C" int sug_interp3_cat_bwd(const float* g, int64_t ldg, int C1, const float* node,
                                   const int32_t* idx3, const float* d3, const float* xyz,
                                   const float* nloc, int B, int N, int S, int C2, float* dnode,
                                   float* dnloc, void* stream) {
  SUG_REQUIRE(g && node && idx3 && d3 && xyz && nloc && dnode && dnloc, "sug_interp3_cat_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S >= 3 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ldg % 4 == 0 &&
                  ldg >= C1 + C2 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)node % 16 == 0),
              "sug_interp3_cat_bwd: bad shape / alignment");
  const size_t sh = (size_t)(S * C2 + S * 3) * sizeof(float);
  SUG_REQUIRE(sh <= 64 * 1024 && B <= 65535, "sug_interp3_cat_bwd: S*C2=%d too large for the LDS accumulator", S * C2);
  int chunks = 512 / B;                      // ~512 workgroups in total
  if (chunks < 1) chunks = 1;
  if (chunks > sug_divup(N, 64)) chunks = sug_divup(N, 64);
  hipLaunchKernelGGL(interp3_cat_bwd_kernel, dim3(chunks, B), dim3(256), sh, (hipStream_t)stream, g, ldg, C1, node,
                     idx3, d3, xyz, nloc, N, S, C2, chunks, dnode, dnloc);
  SUG_LAUNCH_CHECK("sug_interp3_cat_bwd");
  return SUG_OK;
}

// The same gradient without LDS accumulation: scratch rev_off [B,S+1], rev_ent [B,3N] (ints), ddw [B,N,6] (floats);
// dnode / dnloc are written entirely (no zero fill); fixed summation order (sorted reverse lists).
extern "C" int sug_interp3_cat_bwd_lists(const float* g, int64_t ldg, int C1, const float* node, const int32_t* idx3,
                                         const float* d3, const float* xyz, const float* nloc, int B, int N, int S, int C2,
                                         int32_t* rev_off, int32_t* rev_ent, float* ddw, float* dnode, float* dnloc,
                                         void* stream) {
  SUG_REQUIRE(g && node && idx3 && d3 && xyz && nloc && rev_off && rev_ent && ddw && dnode && dnloc,
              "sug_interp3_cat_bwd_lists: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S >= 3 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ldg % 4 == 0 &&
                  ldg >= C1 + C2 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)node % 16 == 0) && ((uintptr_t)dnode % 16 == 0),
              "sug_interp3_cat_bwd_lists: bad shape / alignment");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = sug_reverse_lists(idx3, B, 3 * N, S, 1, rev_off, rev_ent, st)) return rc;
  const int64_t BN = (int64_t)B * N, BS = (int64_t)B * S;
  hipLaunchKernelGGL(interp3_bwd_point_kernel, dim3(sug_divup(BN, 16)), dim3(256), 0, st, g, ldg, C1, node, idx3, d3, N, S, C2,
                     BN, ddw);
  SUG_LAUNCH_CHECK("sug_interp3_cat_bwd_lists(point)");
  hipLaunchKernelGGL(interp3_bwd_node_kernel, dim3(sug_divup(BS, 16)), dim3(256), 0, st, g, ldg, C1, ddw, rev_off, rev_ent, xyz,
                     nloc, N, S, C2, BS, dnode, dnloc);
  SUG_LAUNCH_CHECK("sug_interp3_cat_bwd_lists(node)");
  return SUG_OK;
}
