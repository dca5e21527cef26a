/*
 * sug_amd.h -- C ABI of libsug_amd.so: MI355X (gfx950) kernels for SUG's
 * point-cloud encoder + MMD alignment hot path.
 *
 * Conventions (mirroring the reference's only native interface, the pybind
 * module of model/pointnet2/src/pointnet2_api.cpp:10-24):
 *   - the caller owns every buffer (device pointers), pre-allocates all outputs
 *     and scratch; kernels write in place and retain nothing;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     re-entrant, holds no global state, never synchronises, never allocates (the only process-wide note kept is, per device, that a
 *     kernel's dynamic-LDS limit has been raised);
 *   - return 0 on success or a negative SUG_ERR_* code (never exit());
 *     sug_last_error() returns a thread-local message for the last failure;
 *   - feature tensors are point-major rows ("channel-last"): element (b,n,c) of
 *     a [B,N,C] tensor lives at base[(b*N+n)*ld + c]; `ld` (row stride, in
 *     elements, >= C) lets a caller write/read a column slice of a wider
 *     buffer.  xyz tensors are dense [B,N,3].  Indices are int32.
 *   - all arithmetic is IEEE fp32 with a fixed operation order (stated per
 *     function) so that index results are bit-reproducible against the CPU
 *     reference path; nothing is compiled with fast-math or fp contraction.
 */
#ifndef SUG_AMD_H
#define SUG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUG_OK 0
#define SUG_ERR_ARG (-1)          /* bad argument / unsupported size        */
#define SUG_ERR_LAUNCH (-2)       /* hipLaunch failed                       */
/* Per-channel sums (BN statistics, BN gradients) are reduced without atomics so that
 * results are bit-reproducible run to run: kernels write one fp32 partial row per
 * workgroup into a caller-provided workspace `ws` of SUG_STATS_BLOCKS * 2*C floats,
 * then an ordered fp64 pass fills the 2*C-double result (no zeroing needed).
 * The BatchNorm entry points (sug_col_stats_bn, sug_bn_act_rows_fwd, sug_pointmlp_max_layer_fwd,
 * sug_sa_first_fwd, sug_bn_act_pool_layer_fwd) take the batch statistics about a pivot row (one value of
 * the group's data per channel: sum(x - p), sum((x - p)^2) in fp32, mean = p + S1/n and
 * var = S2/n - (S1/n)^2 in fp64), so that data with |mean| >> std keeps its variance, like the two-pass /
 * Welford forms of nn.BatchNorm; the pivot rows live in the last 8 rows of the same workspace. */
#define SUG_STATS_BLOCKS 1024

const char* sug_last_error(void);
/* ABI version of the loaded library (bumped when a signature changes; 3: sug_adam_step_capturable gained lr_dev,
 * round-3 entry points; 4: sug_chamfer / sug_node_offset_bwd became reproducible -- sug_chamfer takes a workspace; 5: sug_ce_pair_* take ignore_index, lse has 2M + 1 entries; sug_pointmlp_max_layer_fwd_xf and sug_col_stats_bn_grouped added; 6: sug_ptran_fused_fwd / sug_ptran_fused_supported added, sug_group_max_bwd fails instead of changing its summation order when the LDS opt-in is refused; 7: sug_adam_chain_step, sug_edge_weight_split_multi, sug_soft_mmd_multi_fwd / _bwd, sug_sda_prob_weights_multi, sug_chamfer_weights and sug_bn_replay_multi added, sug_bn_act_pool_* take the row stride ld_pool of the pooled outputs / their gradients).  A binding checks sug_abi_version() == SUG_ABI_VERSION of the header it was written against. */
#define SUG_ABI_VERSION 7
int sug_abi_version(void);

/* ---- kNN graph ----------------------------------------------------------
 * replaces knn(), model/model_utils.py:178-185 (the [B,N,N] matrix + topk).
 * score(i,j) = (-|x_j|^2 - (-2 * <x_i,x_j>)) - |x_i|^2, <.,.> an ascending-c
 * fma chain, |.|^2 a sequential sum of separately rounded squares; the k
 * largest scores per i, descending, ties -> lowest j first.  Requires
 * N >= k, 1 <= k <= 32. x: [B,N,C] rows (ld = ldx). idx: [B,N,k]. */
int sug_knn(const float* x, int64_t ldx, int B, int N, int C, int k,
            int32_t* idx, void* stream);

/* Reverse neighbour lists of a kNN graph (needed by the EdgeConv backward):
 * for every destination point m the entries e = n*k + j with idx[b,n,j] == m,
 * ascending in e.  rev_off: [B,N+1] (exclusive prefix, per cloud), rev_ent:
 * [B,N*k].  No reference counterpart (autograd's index_put_ does this
 * implicitly, model/model_utils.py:204). */
int sug_knn_reverse(const int32_t* idx, int B, int N, int k,
                    int32_t* rev_off, int32_t* rev_ent, void* stream);

/* ---- farthest point sampling -------------------------------------------
 * replaces farthest_point_sample(), model/point_utils.py:5-26,
 * model/pointnet2_utils.py:60-81, model/PTran_utils.py:53-73, and
 * furthest_point_sampling_wrapper (model/pointnet2/src/sampling.cpp:38-39).
 * d = ((dx*dx + dy*dy) + dz*dz); running min; arg-max ties -> lowest index.
 * start[b] is the first centroid (the caller draws it the way the reference
 * does: torch.randint on the CPU generator).  xyz [B,N,3], out [B,npoint].
 * N <= 8192. */
int sug_fps(const float* xyz, const int32_t* start, int B, int N, int npoint,
            int32_t* out, void* stream);

/* ---- ball query -----------------------------------------------------------
 * replaces query_ball_point(radius != None), model/point_utils.py:99-106,
 * model/pointnet2_utils.py:97-103 and ball_query_wrapper_fast
 * (model/pointnet2/src/ball_query.cpp:14-15) with the *torch path's* rule:
 * d = ((-2*<q,p>) + |q|^2) + |p|^2; keep !(d > r2); first nsample hits in
 * ascending index order, short rows padded with the first hit, rows with no
 * hit filled with N.  xyz [B,N,3], query [B,S,3], out [B,S,nsample]. */
int sug_ball_query(const float* xyz, const float* query, int B, int N, int S,
                   float r2, int nsample, int32_t* out, void* stream);

/* ---- k nearest candidates of a few queries (full-sort semantics) ---------
 * replaces query_ball_point(radius=None), model/point_utils.py:107-108
 * (sort of the [B,S,N] distances, first k): ascending d (same formula as the
 * ball query), ties -> lowest index.  k <= 64, N <= 4096.
 * dist_out may be NULL; else [B,S,k]. */
int sug_knn_query(const float* xyz, const float* query, int B, int N, int S,
                  int k, int32_t* idx_out, float* dist_out, void* stream);
/* Same selection on the direct-form distance sum((q - p)^2) of the Point Transformer path:
 * `square_distance(...).argsort()[:, :, :k]`, Ptran_transformer.py:32-33, PTran_utils.py:117-119. */
int sug_knn_query_direct(const float* xyz, const float* query, int B, int N, int S, int k,
                         int32_t* idx_out, float* dist_out, void* stream);

/* ---- 3 nearest of few candidates for many queries --------------------------
 * replaces the sort + [:3] of upsample_inter(), model/point_utils.py:153-155
 * and three_nn_wrapper_fast (model/pointnet2/src/interpolate.cpp:14-15).
 * query [B,N,3] (the dense cloud), cand [B,S,3] (the nodes), S <= 2048.
 * idx3/dist3: [B,N,3], ascending d, ties -> lowest index; d is the raw
 * expanded-form value (the 1e-10 clamp is the caller's). */
int sug_three_nn(const float* query, const float* cand, int B, int N, int S,
                 int32_t* idx3, float* dist3, void* stream);

/* dst[r][0..C) = src[r][0..C), r < rows, row strides lds / ldd in floats (C, lds, ldd multiples of 4; 16-byte aligned):
 * a dense tensor into a column slice of a wider row buffer (the torch.cat of DGCNN.forward, model/Model.py:111) or back. */
int sug_copy_rows2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C, void* stream);

/* ---- row gather / scatter (index_points) ------------------------------------
 * replaces index_points(), model/point_utils.py:60-83,
 * model/pointnet2_utils.py:41-57 and gather_points/group_points wrappers
 * (model/pointnet2/src/pointnet2_api.cpp:13-18).
 * out[b,s,:] = feat[b, idx[b,s], :]; idx [B,S] (S may be npoint*nsample).
 * Out-of-range indices (the reference's zero-hit value N) read as 0. */
int sug_gather_rows(const float* feat, int64_t ldf, const int32_t* idx,
                    int B, int N, int S, int C, float* out, int64_t ldo, void* stream);
/* dfeat[b, idx[b,s], :] += g[b,s,:]  (dfeat must be zero-initialised by the caller) */
int sug_scatter_add_rows(const float* g, int64_t ldg, const int32_t* idx,
                         int B, int N, int S, int C, float* dfeat, int64_t ldf, void* stream);
/* The same sum in ONE fixed order (reproducible bit for bit): sorted reverse lists of idx (which entries name point n,
 * ascending), then dfeat[b,n,:] = sum over n's entries of g -- every row of dfeat is WRITTEN (no zero fill by the caller).
 * ws: sug_scatter_rows_workspace(B, N, S) int32.  The reverse lists are built in LDS: sug_scatter_rows_ordered_supported()
 * is 0 when S entries per cloud do not fit (callers then use sug_scatter_add_rows: atomics, unordered, as the reference's
 * index_points backward). */
int64_t sug_scatter_rows_workspace(int B, int N, int S);
int sug_scatter_rows_ordered_supported(int B, int N, int S);
int sug_scatter_rows_ordered(const float* g, int64_t ldg, const int32_t* idx, int B, int N, int S, int C,
                             float* dfeat, int64_t ldf, int32_t* ws, void* stream);

/* out[b,s,c] = max_j feat[b, idx[b,s,j], c], arg[b,s,c] = the winning point index
 * (replaces index_points + torch.max(dim=-1), model/model_utils.py:122-123). */
int sug_group_max(const float* feat, int64_t ldf, const int32_t* idx, int B, int N, int S,
                  int ns, int C, float* out, int32_t* arg, void* stream);
/* Backward: dfeat[b, arg[b,s,c], c] += g[b,s,c].  CONTRACT: dfeat [B, N, ldf] must be ZERO-FILLED by the caller; it is not an
 * accumulate-into entry point.  (S <= 64: a fixed-order kernel that STORES the sums of the touched entries; otherwise float
 * atomics that ADD -- identical on a zeroed buffer, unspecified on any other.) */
int sug_group_max_bwd(const float* g, const int32_t* arg, int B, int N, int S, int C,
                      float* dfeat, int64_t ldf, void* stream);

/* ---- EdgeConv: neighbour gather + BN statistics + max over k -----------------
 * replaces get_graph_feature + conv_2d + max(dim=-1),
 * model/model_utils.py:188-210, :8-32 and model/Model.py:88-109, using
 * W.[x_j - x_i ; x_i] = W1.x_j + (W2-W1).x_i: the caller provides
 * PQ[b,n,:] = [P | Q] = x[b,n,:] . [W1 ; W2-W1]^T (one dense GEMM, [B*N, 2*Co],
 * row stride ldpq) and this kernel forms y[b,n,j,c] = P[b,idx[b,n,j],c] + Q[b,n,c]
 * on the fly, never materialising the k-expanded tensor.
 * Outputs: z[b,n,c]   = max_j y (gamma[c] >= 0) or min_j y (gamma[c] < 0) -- the
 *                       element BN+LeakyReLU (monotone per channel) maps to the max;
 *          arg[b,n,c] = the winning j (uint8);
 *          s1[b,n,c]  = sum_j y (may be NULL; needed for the backward);
 *          stats[0:Co] = sum y, stats[Co:2Co] = sum y^2 over all B*N*k values (fp64);
 *          ws: workspace, SUG_STATS_BLOCKS*2*Co floats.
 * Co % 4 == 0, Co <= 1024, k <= 255. */
int sug_edgeconv_fwd(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                     int B, int N, int k, int Co, float* z, uint8_t* arg, float* s1,
                     double* stats, float* ws, void* stream);

/* Train-mode BatchNorm bookkeeping for a channel-last tensor: from fp64 sums of
 * `count` values per channel produce mean/rstd and the folded affine
 * scale = gamma*rstd, shift = beta - mean*scale; update running_mean/var the
 * way nn.BatchNorm2d does (momentum, unbiased running variance).
 * coef: [5,C] = scale, shift, mean, rstd, unbiased variance.  running_* may be NULL. */
int sug_bn_finalize(const double* stats, const float* gamma, const float* beta, int C,
                    double count, float eps, float momentum, float* running_mean,
                    float* running_var, float* coef, void* stream);

/* Fused forms used by the layer entry points: the producer's per-workgroup partial statistics are
 * folded (same fixed order) and turned into coef / running-buffer updates by one kernel.
 * sug_edgeconv_fwd_bn = sug_edgeconv_fwd + sug_bn_finalize (count = B*N*k);
 * sug_col_stats_bn    = sug_col_stats   + sug_bn_finalize (count = rows). */
int sug_edgeconv_fwd_bn(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                        const float* beta, int B, int N, int k, int Co, float eps, float momentum,
                        float* running_mean, float* running_var, float* z, uint8_t* arg, float* s1,
                        float* coef, float* ws, void* stream);
int sug_col_stats_bn(const float* y, int64_t ldy, int64_t rows, int C, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, float* coef,
                     float* ws, void* stream);

/* sug_col_stats_bn for `groups` equal consecutive row blocks of y [rows, C] (the domain groups of a paired batch): coef
 * [groups][5][C], running buffers updated in group order; one launch pair for all groups where the layout allows. */
int sug_col_stats_bn_grouped(const float* y, int64_t ldy, int64_t rows, int C, int groups, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                             float* coef, float* ws, void* stream);

/* Replay of the running-statistics update for G more train-mode forwards over batches whose
 * statistics are already in coef [G,5,C] (rows 2 and 4 of each group), in group order, with
 * sug_bn_finalize's arithmetic: what nn.BatchNorm would do when the encoder prefix is evaluated
 * again on the same batch (model/Model.py:88-92 run once per forward call). */
int sug_bn_replay(const float* coef, int G, int C, float momentum, float* running_mean, float* running_var,
                  void* stream);
/* The same for n <= 16 layers in one launch (HOST arrays of n entries; a layer must not appear twice). */
int sug_bn_replay_multi(int n, const void* const* coef, const int32_t* G, const int32_t* C, const float* momentum,
                        void* const* running_mean, void* const* running_var, void* stream);

/* out[r,c] = act(scale[c]*z[r,c] + shift[c]), act = LeakyReLU(slope) (slope 0: ReLU,
 * slope 1: identity).  rows = B*N.  */
int sug_affine_act(const float* z, int64_t ldz, const float* coef, int64_t rows, int C,
                   float slope, float* out, int64_t ldo, void* stream);

/* Column sums for BN over rows of a channel-last tensor: stats[0:C] = sum,
 * stats[C:2C] = sum of squares (fp64).  Used for per-point conv_2d layers
 * (model/model_utils.py:8-32 on [B,C,N,1]).  ws: SUG_STATS_BLOCKS*2*C floats. */
int sug_col_stats(const float* y, int64_t ldy, int64_t rows, int C, double* stats, float* ws,
                  void* stream);

/* EdgeConv backward, step 1: G = gout * act'(scale*z+shift);
 * a[b,n,c] = scale[c]*G;  red[0:Co] = sum G, red[Co:2Co] = sum G*(z-mean)*rstd (fp64).
 * ws: SUG_STATS_BLOCKS*2*Co floats. */
int sug_edgeconv_bwd_reduce(const float* gout, int64_t ldg, const float* z, const float* coef,
                            int64_t rows, int Co, float slope, float* a, double* red, float* ws,
                            void* stream);
/* EdgeConv backward, step 2: dPQ [B*N, 2*Co] (row stride lddpq) from a, arg, s1, PQ,
 * the reverse lists and the reduced sums (exact train-mode BN gradient):
 *  dQ[n,c] = a[n,c] - (scale/M)*(k*dbeta + rstd*dgamma*(s1[n,c] - k*mean))
 *  dP[m,c] = sum_{(n,j) in rev(m)} (a[n,c]*[arg[n,c]==j] - (scale/M)*rstd*dgamma*Q[n,c])
 *            - (scale/M)*cnt[m]*(dbeta + rstd*dgamma*(P[m,c]-mean)),   M = B*N*k. */
int sug_edgeconv_bwd_scatter(const float* a, const uint8_t* arg, const float* s1,
                             const float* pq, int64_t ldpq, const int32_t* rev_off,
                             const int32_t* rev_ent, const float* coef, const double* red,
                             int B, int N, int k, int Co, float* dpq, int64_t lddpq, void* stream);

/* ---- weight gradient of a per-point linear layer ----------------------------------------------
 * dw[M,N] = g^T . x, g [R,M] (row stride ldg), x [R,N] (row stride ldx): the backward of every
 * 1x1 conv of the encoders (conv_2d / Conv1d, model/model_utils.py:13, model/Model.py:68-70)
 * w.r.t. its weight, R = B*N rows.  Split over row chunks, fp32 MFMA, ordered combine
 * (bit-reproducible).  ws: sug_linear_dw_workspace(R,M,N) floats. */
int64_t sug_linear_dw_workspace(int64_t R, int M, int N);
int sug_linear_dw(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int N,
                  float* dw, float* ws, void* stream);
/* Ordered fold of nchunk split-K partials part[nchunk][MN] -> dw[MN] (fp64 accumulation in a fixed order; no atomics,
 * no memset): the second half of sug_linear_dw, for partials a batched library GEMM over row chunks produced. */
int sug_linear_dw_fold(const float* part, int nchunk, int64_t MN, float* dw, void* stream);
/* The same launch also forms the bias gradient db[M] = column sums of g (the Conv bias, nn.Conv2d(bias=True) of
 * model/pointnet2_utils.py:172 and the nn.Linear biases of model/Ptran_transformer.py:17-33) from the g elements
 * it already holds, instead of a second pass over g.  db == NULL: sug_linear_dw. */
int sug_linear_dw_bias(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int N,
                       float* dw, float* db, float* ws, void* stream);

/* out[c] = sign * sum over rows of x[r, c] (fp32 sums of fp32 (dtype 0) or fp16 (1) rows, row stride ld): the bias
 * gradients autograd forms with sum(dim=0) / sum_to_size (nn.Linear / Conv biases, model/pointnet2_utils.py:172,
 * model/Ptran_transformer.py:17-33), without the memset torch's reduction issues for tall shapes.
 * ws: sug_colsum_workspace(rows, C) floats. */
int64_t sug_colsum_workspace(int64_t rows, int C);
int sug_colsum(const void* x, int64_t ld, int64_t rows, int C, int dtype, float sign, float* out, float* ws, void* stream);

/* Channel-attention gate of CALayer (model/Model.py:28-34): out = x * sigmoid(z) + x over n elements, z = the
 * output of the second 1x1 conv; backward dx = g * sigmoid(z) + g, dz = g * x * sigmoid'(z). */
int sug_gate_fwd(const float* x, const float* z, int64_t n, float* out, void* stream);
int sug_gate_bwd(const float* g, const float* x, const float* z, int64_t n, float* dx, float* dz, void* stream);
/* The gate followed by CALayer's BatchNorm1d (model/Model.py:442-449: attention_s / attention_t on [M, 4096] node
 * features, M = the clouds of one domain): out = BN(x * sigmoid(z) + x), train mode = batch statistics over the M rows
 * (two-pass) with the running buffers updated (null: not tracked), eval mode = running statistics; stat [2, C] = mean |
 * invstd for the backward, which returns dx, dz of the gate and dgamma, dbeta [C].  One launch each way. */
int sug_gate_bn_fwd(const float* x, const float* z, int M, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int training, float eps, float momentum, float* out,
                    float* stat, void* stream);
int sug_gate_bn_bwd(const float* g, const float* x, const float* z, int M, int C, const float* gamma, const float* stat,
                    int training, float* dx, float* dz, float* dgamma, float* dbeta, void* stream);

/* ---- Channel attention (CALayer, model/Model.py:16-34) for up to two attention layers per launch ------------------
 * v [M, C] -> h = relu(v . W0^T + b0) [M, Hd] -> z = h . W2^T + b2 [M, C] (the two 1x1 Conv2d on a 1x1 map; the gate
 * v * sigmoid(z) + v and the BatchNorm1d behind it are sug_gate_bn_fwd / _bwd).  Net_MDA has two such layers
 * (attention_s on the source rows, attention_t on the target rows of a paired batch): x [layers * M, C] holds the rows of
 * layer 0 first; W0 / b0 / W2 / b2 (and the gradient outputs) are HOST arrays of `layers` device pointers.
 * M <= 64, Hd = 512, C % 512 == 0 (C = 4096).  Scratch: hp [layers][C/512][M][Hd], dhp [layers][C/32][M][Hd].
 * Forward writes h [layers][M][Hd] and z [layers*M, C]; backward takes dz (from sug_gate_bn_bwd) and dxg (the gate's
 * own input gradient, added to dx; may be null) and writes dW0 [Hd,C], db0 [Hd], dW2 [C,Hd], db2 [C] per layer, dh
 * [layers][M][Hd] (scratch) and dx [layers*M, C] (row stride lddx).  Fixed summation order, no atomics. */
int sug_calayer_supported(int layers, int M, int C, int Hd);
int sug_calayer_fwd(int layers, const float* x, int64_t ldx, int M, int C, int Hd, const float* const* W0,
                    const float* const* b0, const float* const* W2, const float* const* b2, float* hp, float* h, float* z,
                    void* stream);
int sug_calayer_bwd(int layers, const float* x, int64_t ldx, int M, int C, int Hd, const float* const* W0,
                    const float* const* W2, const float* h, const float* dz, const float* dxg, float* const* dW0,
                    float* const* db0, float* const* dW2, float* const* db2, float* dhp, float* dh, float* dx, int64_t lddx,
                    void* stream);

/* ---- Classifier heads: Linear layers with M <= 128 rows, one launch per layer for up to two heads ----------------
 * Pointnet_c (model/Model.py:412-449): fc_layer(1024, 512) -> Dropout -> fc_layer(512, 256) [= mid feature] -> Dropout ->
 * Linear(256, num_class), fc_layer = Linear -> LayerNorm -> LeakyReLU(0.2) / ReLU (model/model_utils.py:35-57); Net_MDA
 * runs two such heads (c1, c2) on the same pooled feature.  All per-head operands are HOST arrays of `heads` device
 * pointers (entries / whole arrays may be null where noted).  Every sum runs in a fixed order.  8 launches replace the
 * ~47 library / elementwise launches of the two heads per training step.
 *
 * sug_head_linear_fwd: z[h] [M, No] = pro(in[h]) . W[h]^T + bias[h].  pro != 0: in[h] is the PREVIOUS layer's
 *   pre-LayerNorm output and pro(v) = Dropout(act(LayerNorm(v; gamma[h], beta[h]))), dropout from uniform randoms u[h]
 *   [M, K] (keep when u >= p_drop, scale 1/(1-p_drop); u null: no dropout); side outputs stats[h] [M, 2] = mean | rstd of
 *   the rows of in[h] and act[h] [M, K] = the activation before the dropout (null: not written).
 *   K % 256 == 0 (pro: K <= 1024), No <= 32 or 256 or 512, M <= 128 (sug_head_linear_supported).
 * sug_head_ln_bwd: dz[h] [M, No] = gradient of a layer's Linear output z[h] from gup[h] [M, No] (row stride ldg) = the
 *   gradient of Dropout(act(LayerNorm(z))), plus gextra[h] = a gradient of the activation BEFORE the dropout (the mid
 *   feature's; may be null); also dgamma[h], dbeta[h] [No] and db[h] [No] = the column sums of dz (null: none).
 *   No = 256 or 512.
 * sug_head_linear_bwd: from dz[h] [M, No] (row stride ldg): dW[h] [No, K] = dz^T . a, db[h] [No] = column sums of dz
 *   (null: none) and the input gradient dz . W[h] [M, K] (row stride ldda): da[h] per head, or with sum_da != 0 the sum
 *   over the heads into da[0] (the heads share their input).  The layer input a is in[h] itself (pro == 0) or
 *   pro(in[h]) recomputed from stats_in / gamma_in / beta_in / u_in (dropout probability p_drop_in). */
int sug_head_linear_supported(int M, int K, int No, int pro, int epi);
int sug_head_linear_fwd(int heads, const float* const* in, int64_t ldin, const float* const* W, const float* const* bias,
                        float* const* z, const float* const* gamma, const float* const* beta, const float* const* u,
                        float* const* stats, float* const* act, int M, int K, int No, int pro, float slope, float eps,
                        float p_drop, void* stream);
int sug_head_ln_bwd(int heads, const float* const* gup, int64_t ldg, const float* const* z, const float* const* stats,
                    const float* const* gamma, const float* const* beta, const float* const* u,
                    const float* const* gextra, float* const* dz, float* const* dgamma, float* const* dbeta,
                    float* const* db, int M, int No, float slope, float p_drop, void* stream);
int sug_head_linear_bwd(int heads, int sum_da, const float* const* dz, int64_t ldg, const float* const* in, int64_t ldin,
                        const float* const* stats_in, const float* const* gamma_in, const float* const* beta_in,
                        const float* const* u_in, const float* const* W, float* const* dW, float* const* db,
                        float* const* da, int64_t ldda, int M, int K, int No, int pro, float slope, float p_drop_in,
                        void* stream);

/* ---- LayerNorm + (Leaky)ReLU of the FC heads ---------------------------------------------------
 * fc_layer (model/model_utils.py:35-57: nn.Linear -> nn.LayerNorm -> LeakyReLU(0.2) / ReLU) behind the Linear:
 * y = act(LN(x)) over rows of x [rows, C] (C <= 1024), stat [rows, 2] = mean | rstd for the backward.
 * Backward: dx [rows,C], dgamma, dbeta [C] (rows summed in order); ws: rows*C floats. */
int sug_ln_act_fwd(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float slope,
                   float* y, float* stat, void* stream);
int sug_ln_act_bwd(const float* g, const float* x, const float* gamma, const float* beta, const float* stat, int rows,
                   int C, float slope, float* dx, float* dgamma, float* dbeta, float* ws, void* stream);

/* ---- optimizer step ------------------------------------------------------------------------
 * torch.optim.Adam (no amsgrad; L2 weight decay added to the gradient; bias correction; eps
 * outside the square root) over T tensors in one launch per 384 tensors: the update of
 * optimizer_g / optimizer_c / optimizer_dis, train_dg_single_gpu.py:193-203 and :333-335.
 * table: device int64 [T,4] = {param ptr, exp_avg ptr, exp_avg_sq ptr, numel}; block_first:
 * device + host copies of the int32 [T+1] prefix sum of ceil(numel / sug_adam_chunk());
 * grads_host: HOST array of T device pointers (null = tensor without gradient, skipped);
 * bias_corr1 = 1 - beta1^step, bias_corr2 = 1 - beta2^step.  Hyper-parameters are doubles and the
 * derived constants (1 - beta, lr / bias_corr1) are rounded to fp32 once, as torch does. */
int sug_adam_chunk(void);
int sug_adam_step(const int64_t* table, const int32_t* block_first, const int32_t* block_first_host, int T,
                  const void* const* grads_host, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double bias_corr1, double bias_corr2, void* stream);
/* Same update for replay from a hipGraph: the step count is a device int32 (advanced by the call),
 * the bias-correction scalars are derived on the device into scalars_dev[2]; gradient pointers must
 * be stable across replays (graph-private allocations are).  lr_dev (may be null): the learning rate as
 * one device-resident double that overrides `lr`, so that a learning-rate schedule
 * (train_dg_single_gpu.py:194-212) changes a value in memory and ONE captured graph serves every epoch. */
int sug_adam_step_capturable(const int64_t* table, const int32_t* block_first, const int32_t* block_first_host,
                             int T, const void* const* grads_host, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int32_t* step_dev, float* scalars_dev,
                             const double* lr_dev, void* stream);

/* The three optimizer steps of train_dg_single_gpu.py:333-335 in ONE update launch.  optimizer_dis and optimizer_g both
 * own the encoder's parameters (:193-203), so a tensor carries an ordered list of up to SUG_ADAM_CHAIN_SLOTS
 * (exp_avg, exp_avg_sq, bucket) slots; the updates of a tensor are applied one after the other on the value held in
 * registers, each with exactly sug_adam_step's arithmetic: bit-identical to stepping the optimizers in that order, with
 * the parameter and its gradient read and written once.  A bucket = one set of hyper-parameters + one step count
 * (at most SUG_ADAM_CHAIN_BUCKETS).
 * table: device int64 [T,8] = {param, numel, exp_avg0, exp_avg_sq0, exp_avg1, exp_avg_sq1,
 *        nslots | bucket0 << 8 | bucket1 << 16, 0};  block_map: device int32 [blocks,2] = (tensor, chunk) per workgroup of
 *        sug_adam_chain_chunk() elements, tensors in table order;  block_first_host: HOST int32 [T+1] prefix sum of the
 *        tensors' workgroup counts;  grads_host: HOST array of T device pointers (null = skipped);
 * hyper_host: HOST double [nbuckets,6] = {lr, beta1, beta2, eps, weight_decay, t} with t the step count the update
 *        belongs to (1-based; read only when steps_dev is null);
 * steps_dev / scalars_dev (both or neither): device int32 [nbuckets] advanced by the call and float [nbuckets,2] scratch,
 *        for replay from a hipGraph (as sug_adam_step_capturable); lr_dev (may be null): device double [nbuckets]
 *        overriding hyper_host's lr. */
#define SUG_ADAM_CHAIN_SLOTS 2
#define SUG_ADAM_CHAIN_BUCKETS 8
int sug_adam_chain_chunk(void);
int sug_adam_chain_step(const int64_t* table, const int32_t* block_map, const int32_t* block_first_host, int T,
                        const void* const* grads_host, int nbuckets, const double* hyper_host, int32_t* steps_dev,
                        float* scalars_dev, const double* lr_dev, void* stream);

/* ---- SA-node module glue (adapt_layer_off, model/model_utils.py:103-128) --------------------
 * off[b,s,:] = mean_j tanh(proj[b,g_j,:] - proj[b,f,:]) * (loc[b,g_j,:] - loc[b,f,:]),
 * nloc = loc[b,f,:] + off, with f = fidx[b,s], g_j = gidx[b,s,j], proj = fea . W_pred_offset^T
 * ([B,N,3]; :110-119).  Backward: dproj [B,N,3], written entirely by the call (accumulated per cloud in LDS). */
int sug_node_offset_fwd(const float* proj, const float* loc, const int32_t* fidx,
                        const int32_t* gidx, int B, int N, int S, int ns, float* off, float* nloc,
                        void* stream);
int sug_node_offset_bwd(const float* proj, const float* loc, const int32_t* fidx,
                        const int32_t* gidx, const float* goff, int B, int N, int S, int ns,
                        float* dproj, void* stream);
/* out[b,n,:] = [ fea[b,n,0:C1] | sum_t w_t * node[b, idx3[b,n,t], 0:C2] ] with the inverse-distance
 * weights of upsample_inter (model/point_utils.py:156-162) from the 3-NN distances d3.
 * Backward (g = d out): dnode [B,S,C2] and dnloc [B,S,3] (both zero-initialised by the caller,
 * atomics); dnloc carries d3's gradient through d = |xyz - nloc|^2; d fea is g[:, :, 0:C1]. */
int sug_interp3_cat_fwd(const float* fea, int64_t ldf, int C1, const float* node,
                        const int32_t* idx3, const float* d3, int B, int N, int S, int C2,
                        float* out, int64_t ldo, void* stream);
int sug_interp3_cat_bwd(const float* g, int64_t ldg, int C1, const float* node, const int32_t* idx3,
                        const float* d3, const float* xyz, const float* nloc, int B, int N, int S,
                        int C2, float* dnode, float* dnloc, void* stream);
/* The same gradient by destination node over the (sorted) reverse lists of idx3: no accumulation in LDS, fixed
 * summation order; dnode / dnloc are written entirely.  Scratch: rev_off [B,S+1], rev_ent [B,3N] ints, ddw [B,N,6] floats. */
int sug_interp3_cat_bwd_lists(const float* g, int64_t ldg, int C1, const float* node, const int32_t* idx3,
                              const float* d3, const float* xyz, const float* nloc, int B, int N, int S, int C2,
                              int32_t* rev_off, int32_t* rev_ent, float* ddw, float* dnode, float* dnloc,
                              void* stream);

/* ---- BatchNorm(+act) on rows: backward, and the fused DGCNN tail ----------------------------
 * Exact train-mode BN gradient for a per-point layer (conv_2d on [B,C,N,1],
 * model/model_utils.py:8-32): with a = scale*G and red = (sum G, sum G*xhat) from
 * sug_edgeconv_bwd_reduce (z := y, the BN input),
 *   dy = a - (scale/rows) * (dbeta + xhat*dgamma).   a: dense [rows,C]. */
int sug_bn_bwd_apply(const float* a, const float* y, int64_t ldy, const float* coef,
                     const double* red, int64_t rows, int C, float* dy, int64_t lddy, void* stream);

/* BatchNorm1d -> LeakyReLU(slope) -> max over N | mean over N  (model/Model.py:112-116),
 * one read of y [B,N,C]; coef from sug_bn_finalize.  out_max/out_mean [B,C] with row stride ld_pool (ld_pool = 2C and
 * out_mean = out_max + C: the two land in the torch.cat((max, mean), 1) of Model.py:116 directly), arg [B,C] dense =
 * row of the (first) maximum.  ws: workspace of 12*B*C floats (per-chunk partial max/sum/arg). */
int sug_bn_act_pool_fwd(const float* y, int64_t ldy, const float* coef, int B, int N, int C,
                        float slope, float* out_max, float* out_mean, int64_t ld_pool, int32_t* arg, float* ws,
                        void* stream);
/* Backward of the above including the BN statistics terms: dy [B,N,C] (row stride lddy),
 * red[0:C] = dbeta, red[C:2C] = dgamma (fp64).  train = 0: statistics are constants (eval mode).
 * gmax / gmean [B,C] with row stride ld_pool (the two column halves of the gradient of the concatenated feature).
 * ws: SUG_STATS_BLOCKS*2*C floats. B <= SUG_STATS_BLOCKS/4. */
int sug_bn_act_pool_bwd(const float* y, int64_t ldy, const float* coef, const float* gmax,
                        const float* gmean, int64_t ld_pool, const int32_t* arg, int B, int N, int C, float slope,
                        int train, double* red, float* ws, float* dy, int64_t lddy, void* stream);

/* ---- layer-level entry points ---------------------------------------------------------------
 * One call per BatchNorm-carrying layer, chaining the kernels above for `groups` equal contiguous
 * parts of the batch (the domains the reference sends through the network in separate forward
 * calls: statistics, normalisation and running-buffer updates are per part, in order).  coef is
 * [groups,5,C]: written in training mode, read in eval mode (caller fills it from the running
 * buffers).  stats: fp64 [2C] scratch; ws: SUG_STATS_BLOCKS*2*C floats.
 *
 * EdgeConv layer = get_graph_feature + conv_2d + max over k (model_utils.py:188-210, :8-32,
 * Model.py:88-109): z/arg/s1 as sug_edgeconv_fwd, out [B,N,Co] (row stride ldo). */
int sug_edgeconv_layer_fwd(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                           const float* beta, int B, int N, int k, int Co, int groups, int training,
                           float eps, float momentum, float slope, float* running_mean,
                           float* running_var, float* z, uint8_t* arg, float* s1, float* coef,
                           float* out, int64_t ldo, double* stats, float* ws, void* stream);
/* The same layer with the 1x1 convolution inside the gather kernel (edgeconv_fused.hip): x [B*N, Cin] rows (row
 * stride ldx; Cin in {3, 64, 128}), wcat [2Co, Cin] = [W1 ; W2-W1] of the layer's weight W = [W1 | W2] [Co, 2Cin]
 * (sug_edge_weight_split), qbias [Co] = the conv bias or NULL.  A workgroup = (cloud, 16-channel slice) forms its
 * slice of [P | Q] = x.wcat^T with v_mfma_f32_32x32x2_f32 straight into the LDS image its gathers read; [P|Q] is
 * never written to HBM unless pq_out (row stride ldpq >= 2Co) is given -- sug_edgeconv_layer_bwd wants it.  Two
 * launches: MFMA + gather + BatchNorm partial rows, then statistics fold + BatchNorm + LeakyReLU (the fold runs in
 * every workgroup of the second launch; coef [groups,5,Co] and the running buffers are written by one of them).
 * Requires k == 20, Co % 16 == 0, N*64 bytes of LDS (N <= 2400): sug_edgeconv_fused_supported.
 * ws: SUG_STATS_BLOCKS*2*Co floats.  training = 0: coef is an input (running-statistics coefficients).
 * arg, s1 and pq_out serve the backward only and may be NULL (a forward without autograd writes z and out alone). */
int sug_edgeconv_fused_supported(int N, int k, int Cin, int Co);
int sug_edgeconv_fused_layer_fwd(const float* x, int64_t ldx, int Cin, const float* wcat, const float* qbias,
                                 const int32_t* idx, const float* gamma, const float* beta, int B, int N, int k,
                                 int Co, int groups, int training, float eps, float momentum, float slope,
                                 float* running_mean, float* running_var, float* z, uint8_t* arg, float* s1,
                                 float* pq_out, int64_t ldpq, float* coef, float* out, int64_t ldo, float* ws,
                                 void* stream);
/* (All three layer backward entry points take `dgb`: NULL, or fp32 [2C] receiving the BatchNorm parameter
 * gradients dbeta | dgamma summed over the groups -- sug_fold_groups on `red`.)
 * Its backward: reverse neighbour lists (rev_off [B,N+1], rev_ent [B,N*k], scratch), per-group BN
 * sums red [groups+1, 2Co] (row g: dbeta | dgamma; the spare last row must be zero when
 * training = 0), a [B,N,Co] scratch, dpq [B,N,2Co] (row stride lddpq). */
int sug_edgeconv_layer_bwd(const float* gout, int64_t ldg, const float* z, const uint8_t* arg,
                           const float* s1, const float* pq, int64_t ldpq, const int32_t* idx,
                           const float* coef, int B, int N, int k, int Co, int groups, int training,
                           float slope, float* a, double* red, int32_t* rev_off, int32_t* rev_ent,
                           float* dpq, int64_t lddpq, float* ws, float* dgb, void* stream);
/* ---- first layer of a set-abstraction MLP on the neighbour lists (no grouped tensor) ----------------
 * Replaces index_points(xyz, idx) - new_xyz, index_points(points, idx), torch.cat, mlp_convs[0],
 * mlp_bns[0], F.relu of PointNetSetAbstraction.forward (model/pointnet2_utils.py:107-135, 193-198).
 * W.[x_j - c_s ; f_j] + b = P[j] - Q[s] with P [B,N,C] = W.[x ; f] per point (row stride ldp) and
 * Q [B,S,C] = Wx.c - b per centroid, both formed by the caller (two small GEMMs); idx [B,S,ns] are the
 * ball-query lists (values in [0,N)).  Forward: train-mode statistics of y = P[idx] - Q over the B/groups*S*ns
 * rows of each domain group (coef [groups,5,C] as the other BatchNorm entry points; training = 0: coef is
 * an input), Z [B,S,ns,C] = relu(BN(y)).  C in {64, 128}.  ws: SUG_STATS_BLOCKS*2*C floats.
 * Backward: gz [B,S,ns,C] -> dP [B,N,C] (by destination point over the reverse lists of idx: plain stores, no
 * atomics; the order of a point's sum is as unordered as index_points' backward in the reference), dQ [B,S,C];
 * red [groups+1, 2C] doubles (row g: dbeta | dgamma, the spare last row zero for training = 0); scratch:
 * rev_off [B,N+1], rev_ent [B,S*ns] ints, segsum [2,B,S,C] floats; dgb: NULL or fp32 [2C] = red folded over the groups. */
int sug_sa_first_fwd(const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B, int N, int S,
                     int ns, int C, int groups, const float* gamma, const float* beta, int training,
                     float eps, float momentum, float* running_mean, float* running_var, float* coef,
                     float* Z, float* ws, void* stream);
int sug_sa_first_bwd(const float* gz, const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B,
                     int N, int S, int ns, int C, int groups, int training, const float* coef, double* red,
                     int32_t* rev_off, int32_t* rev_ent, float* segsum, float* dP, float* dQ, float* ws, float* dgb,
                     void* stream);
/* The same layer with the coordinate part taken from the difference the reference forms (pointnet2_utils.py:120
 * grouped_xyz - new_xyz): y = Pf[j] + bias + Wx . (xyz[j] - cent[s]), Pf [B,N,C] = Wf . f per point (NULL when the layer has no
 * input features), xyz [B,N,3], cent [B,S,3] (= new_xyz), Wx [C,3] with row stride ldw (the coordinate columns of the conv
 * weight), bias [C] or NULL.  (An alternative to P[j] - Q[s] kept for diagnostics: measured equally accurate against fp64
 * and 1 % slower per step, so the host mirror uses it only with SUG_SA_FIRST_GEO=1.)  Backward: the same dP (= the gradient of Pf AND of Wx . xyz as autograd sees them) and
 * dQ (= -sum_j dy per centroid: the gradient of Wx . cent - bias). */
int sug_sa_first_geo_fwd(const float* Pf, int64_t ldp, const float* xyz, const float* cent, const float* Wx, int ldw,
                         const float* bias, const int32_t* idx, int B, int N, int S, int ns, int C, int groups,
                         const float* gamma, const float* beta, int training, float eps, float momentum,
                         float* running_mean, float* running_var, float* coef, float* Z, float* ws, void* stream);
int sug_sa_first_geo_bwd(const float* gz, const float* Pf, int64_t ldp, const float* xyz, const float* cent, const float* Wx,
                         int ldw, const float* bias, const int32_t* idx, int B, int N, int S, int ns, int C, int groups,
                         int training, const float* coef, double* red, int32_t* rev_off, int32_t* rev_ent, float* segsum,
                         float* dP, float* dQ, float* ws, float* dgb, void* stream);

/* conv_2d / Conv1d + BatchNorm + (Leaky)ReLU on rows (model_utils.py:8-32, pointnet2_utils.py:195-198):
 * out = act(BN(y)), y [rows,C]. */
int sug_bn_act_rows_fwd(const float* y, int64_t ldy, int64_t rows, int C, int groups, const float* gamma,
                        const float* beta, int training, float eps, float momentum, float slope,
                        float* running_mean, float* running_var, float* coef, float* out, int64_t ldo,
                        double* stats, float* ws, void* stream);
/* Backward: a [rows,C] scratch (= dy when training = 0), red [groups,2C], dy [rows,C]; y dense. */
int sug_bn_act_rows_bwd(const float* gout, int64_t ldg, const float* y, int64_t ldy, const float* coef,
                        int64_t rows, int C, int groups, int training, float slope, float* a, double* red,
                        float* dy, float* ws, float* dgb, void* stream);
/* bn5 -> LeakyReLU -> max | mean over the points (Model.py:112-116) per group; ws_pool: 12*(B/groups)*C floats. */
int sug_bn_act_pool_layer_fwd(const float* y, int64_t ldy, int B, int N, int C, int groups,
                              const float* gamma, const float* beta, int training, float eps, float momentum,
                              float slope, float* running_mean, float* running_var, float* coef,
                              float* out_max, float* out_mean, int64_t ld_pool, int32_t* arg, double* stats, float* ws_stats,
                              float* ws_pool, void* stream);
int sug_bn_act_pool_layer_bwd(const float* y, int64_t ldy, const float* coef, const float* gmax,
                              const float* gmean, int64_t ld_pool, const int32_t* arg, int B, int N, int C, int groups,
                              float slope, int training, double* red, float* ws, float* dy, int64_t lddy,
                              float* dgb, void* stream);

/* ---- per-point MLP layer fused with the max over a group of rows ------------------------------
 * replaces `conv_2d(K, Co)` (1x1 Conv2d + bias -> BatchNorm2d -> ReLU) followed by the max over the
 * points of a cloud: Pointnet_g.conv5 + torch.max (model/Model.py:274-279), transform_net.conv2d3 +
 * maxpool (model/model_utils.py:72-79), Pointnet_cls (model/model_pointnet.py:36-48); and the last
 * layer of a set-abstraction MLP + max over nsample (model/pointnet2_utils.py:193-207).
 * y = x . w^T + bias ([rows,K] x [Co,K]^T, an ascending-k fp32 fma chain on the matrix pipe) is never
 * stored.  Rows form segments of `seg` consecutive rows (a cloud's N points, or a group's nsample
 * points); per (segment, channel) the kernel keeps zext = max_rows y (gamma >= 0) or min_rows y
 * (gamma < 0) -- the element BN + (Leaky)ReLU maps to the max -- and arg = its row inside the segment
 * (lowest row on ties), and it accumulates the BatchNorm sums of y as per-row-block partial rows in
 * ws ([*nblk][2*Co]: sum | sum of squares; *nblk <= SUG_STATS_BLOCKS; ws = SUG_STATS_BLOCKS*2*Co floats: when the
 * grid would leave CUs idle, a segment of >= 1024 rows is split over up to 8 workgroups -- more partial rows, and the
 * rows of ws behind them serve as scratch for the partial extremes, combined in part order by a second small launch:
 * same zext / arg bit for bit).
 * K in {64, 128}; Co % 128 == 0; seg % 32 == 0; rows % seg == 0; x, w 16-byte aligned, ldx % 4 == 0. */
int sug_pointmlp_max_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias,
                         const float* gamma, int Co, int seg, float* zext, int32_t* arg, float* ws, int* nblk,
                         void* stream);
/* y[r,:] = x[r,:] . w^T (+ bias) on the same fp32 MFMA pipeline, y stored ([rows, Co], row stride ldy): the
 * per-point 1x1 convolution itself (conv_2d, model/model_utils.py:8-32) where no reduction follows, e.g. the
 * EdgeConv operand PQ = x . [W1 ; W2-W1]^T.  Same shape limits as sug_pointmlp_max_fwd (rows arbitrary). */
int sug_rows_gemm(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias, int Co,
                  float* y, int64_t ldy, void* stream);
/* Layer entry point: the above per domain group, then the BatchNorm coefficients (training: batch
 * statistics + running-buffer update; eval: coef read, caller fills it) and
 * out[s,c] = LeakyReLU_slope(scale[c]*zext[s,c] + shift[c]), out [rows/seg, Co] (row stride ldo).
 * coef [groups,5,Co]; ws SUG_STATS_BLOCKS*2*Co floats. */
int sug_pointmlp_max_layer_fwd(const float* x, int64_t ldx, int64_t rows, int K, const float* w, const float* bias,
                               const float* gamma, const float* beta, int Co, int seg, int groups, int training,
                               float eps, float momentum, float slope, float* running_mean, float* running_var,
                               float* zext, int32_t* arg, float* coef, float* out, int64_t ldo, float* ws,
                               void* stream);
/* The same layer fed with the PRE-activation rows y [rows, K] of the layer in front (round 5; model/pointnet2_utils.py:
 * 193-207, layers i and i+1 of a set-abstraction MLP): that layer's BatchNorm + (Leaky)ReLU -- xcoef [groups][5][K] as
 * sug_col_stats_bn / sug_bn_finalize write it (rows 0 / 1 = scale / shift), slope xslope -- is applied to the x operand on
 * its way into LDS.  zout [rows, ldz] (may be null): the activated rows z = act(scale * y + shift), written once, for a
 * backward (sug_pointmlp_max_bwd_sparse takes z as its x) or for a caller that needs z itself.  Replaces
 * sug_affine_act on [rows, K] + this kernel's read of its output. */
int sug_pointmlp_max_layer_fwd_xf(const float* y, int64_t ldy, int64_t rows, int K, const float* xcoef, float xslope,
                                  float* zout, int64_t ldz, const float* w, const float* bias, const float* gamma,
                                  const float* beta, int Co, int seg, int groups, int training, float eps, float momentum,
                                  float slope, float* running_mean, float* running_var, float* zext, int32_t* arg,
                                  float* coef, float* out, int64_t ldo, float* ws, void* stream);
/* Backward, the terms that follow the winning rows n*(s,c) = s*seg + arg[s,c], with
 * a[s,c] = scale[c] * gout[s,c] * act' (sug_edgeconv_bwd_reduce on the [rows/seg, Co] tensors):
 *   dx[n*(s,c), :] += a[s,c] * w[c,:]   (dx holds the dense BatchNorm-statistics term -(x.A + v) or zeros)
 *   dw[c,:]         = sum_s a[s,c] * x[n*(s,c), :]
 * both in a fixed summation order (bit-reproducible).  ws: sug_pointmlp_max_bwd_workspace floats. */
/* The dense (rank-K) BatchNorm-statistics terms of the same backward, dy = a_full - k1 - k2*y over ALL rows:
 *   dx = ... - (x.A + v),  A = W^T diag(k2) W,  v = (k1 + k2*b).W;   dW = ... - kb (x) sum(x) - diag(k2) W.(X^T X)
 * _coef: from one group's coefficients coef [5,Co] and folded sums red [2Co] (dbeta | dgamma) over `rows` rows:
 *   k2 = scale/rows*rstd*dgamma, kb = scale/rows*(dbeta - mean*rstd*dgamma) + k2*bias (fp64), and the operands
 *   nkb = -kb [Co], nwk = -diag(k2).W [Co,K] of the two small library products -A = nwk^T.W, -v = nkb.W.
 * _dwfix: dw += dws - (kb (x) sx + diag(k2) W.XtX) with XtX [K,K], sx [K] from sug_linear_dw_bias(x, x)
 *   (with_stats = 0, eval mode: dw += dws). */
int sug_pointmlp_max_bwd_coef(const float* coef, const double* red, const float* bias, const float* w, int64_t rows,
                              int K, int Co, float* kb, float* k2, float* nkb, float* nwk, void* stream);
int sug_pointmlp_max_bwd_dwfix(float* dw, const float* dws, const float* kb, const float* k2, const float* w,
                               const float* xtx, const float* sx, int K, int Co, int with_stats, void* stream);
int64_t sug_pointmlp_max_bwd_workspace(int64_t rows, int K, int Co, int seg);
int sug_pointmlp_max_bwd_sparse(const float* a, const int32_t* arg, const float* x, int64_t ldx, const float* w,
                                int64_t rows, int K, int Co, int seg, float* dx, int64_t lddx, float* dw,
                                float* ws, void* stream);

/* ---- Point Transformer vector attention (BASELINE config 5) --------------------------------------
 * The memory-bound parts of TransformerBlock.forward (model/Ptran_transformer.py:32-45) around the three
 * 512 x 512 linears of the k-expanded rows (library GEMMs: fp32, or fp16 MFMA with fp32 accumulation):
 * rows r = (b*n + i)*k + j, d = 512 channels contiguous, nbr [B,n,k] = neighbour index inside the cloud
 * (sug_knn_query_direct).  dtype: 0 = fp32, 1 = fp16 for the k-expanded tensors T0 / delta / U / L and their
 * gradients (void*); q, K, V [B,n,512], xyz [B,n,3], weights and all reductions are fp32.  k <= 16.
 *   pos1: T0[r,:] = relu(w1 . (xyz_i - xyz_nbr) + b1), w1 [512,3]                    (fc_delta[0] + ReLU, :39)
 *   qk:   U[r,:]  = q[i,:] - K[nbr,:] + delta[r,:]                                    (input of fc_gamma, :41)
 *   attn: mixed[i,:] = sum_j softmax_j(L[r,:]*scale) * (V[nbr,:] + delta[r,:]); mx / sm [B,n,512] = the
 *         per-channel max and sum of exponentials, kept for the backward               (:42-44)
 * Backward: rev_off / rev_ent = sug_knn_reverse(nbr) (dK and dV are gathered over reverse neighbour lists:
 * no atomics).  sug_ptran_attn_bwd: g = d mixed, mixed = the forward's output (its product with g is the softmax
 * backward's row term, so the rows are visited once) -> dlogits, da = the gradient of delta through (v + delta),
 * dv [B,n,512].  sug_ptran_qk_bwd: du = dU; da is read and overwritten with d delta = du + da; dq, dk
 * [B,n,512].  sug_ptran_pos1_bwd: g = dT0 -> dw1 [512,3], db1 [512]; ws: 1024*4*512 floats.
 * attn_bwd and qk_bwd also return the column sums (the bias gradients the caller needs next: model/Ptran_transformer.py's
 * nn.Linear biases of fc_gamma / fc_delta) of the [B*n*k,512] gradient they write - dlogits for attn, d delta for
 * qk - from the same pass instead of one more read of that tensor: db [512] (NULL = skip), ws =
 * sug_ptran_colsum_workspace(B*n) floats.  sug_ptran_relu_bwd_db: in place G <- G*[T1>0] over `rows` rows (the ReLU
 * inside fc_gamma), db [512] = its column sums, ws = sug_ptran_colsum_workspace(rows) floats. */
int sug_ptran_pos1_fwd(const float* xyz, const int32_t* nbr, const float* w1, const float* b1, int B, int n, int k,
                       int d, int dtype, void* out, void* stream);
int sug_ptran_pos1_bwd(const void* g, const float* xyz, const int32_t* nbr, const float* w1, const float* b1, int B,
                       int n, int k, int d, int dtype, float* dw1, float* db1, float* ws, void* stream);
int sug_ptran_qk_fwd(const float* q, const float* kf, const void* delta, const int32_t* nbr, int B, int n, int k, int d,
                     int dtype, void* out, void* stream);
int sug_ptran_qk_bwd(const void* du, void* da, const int32_t* rev_off, const int32_t* rev_ent, int B, int n, int k, int d,
                     int dtype, float* dq, float* dk, float* db, float* ws, void* stream);
int sug_ptran_attn_fwd(const void* logits, const void* delta, const float* vf, const int32_t* nbr, int B, int n, int k,
                       int d, int dtype, float scale, float* mixed, float* mx, float* sm, void* stream);
int sug_ptran_attn_bwd(const float* g, const float* mixed, const void* logits, const void* delta, const float* vf,
                       const int32_t* nbr, const float* mx, const float* sm, const int32_t* rev_off, const int32_t* rev_ent,
                       int B, int n, int k, int d, int dtype, float scale, void* dlogits, void* da, float* dv, float* db,
                       float* ws, void* stream);
/* The fp16 forward of the vector attention as ONE kernel on the matrix cores (csrc/ptran_fused.hip; round 6): pos1, the three
 * 512 x 512 linears of fc_delta[2] / fc_gamma[0] / fc_gamma[2] (v_mfma_f32_32x32x16_f16, activations resident in LDS, weights
 * streamed from L2), U = (q - K_nbr) + delta, the softmax over the 16 neighbours and the weighted sum of V_nbr + delta --
 * model/Ptran_transformer.py:39-44 in the 16-bit mode of BASELINE config 5.  w2 / b2 / wg1 / bg1 / wg2 / bg2: fp16 ([512,512]
 * row-major = nn.Linear.weight, [512]); q / kf / vf / xyz / w1 / b1 fp32.  delta [B n 16, 512] fp16 is always written
 * (the kernel reads it back for the weighted sum); save = 1 additionally writes T0, U, T1 and the logits (what
 * sug_ptran_*_bwd read), save = 0 leaves those pointers unused (NULL allowed).  mixed / mx / sm [B n, 512] fp32 as
 * sug_ptran_attn_fwd.  sug_ptran_fused_supported: 1 for d = 512, k = 16 and B*n a multiple of 8. */
int sug_ptran_fused_supported(int B, int n, int k, int d);
int sug_ptran_fused_fwd(const float* xyz, const int32_t* nbr, const float* q, const float* kf, const float* vf,
                        const float* w1, const float* b1, const void* w2, const void* b2, const void* wg1, const void* bg1,
                        const void* wg2, const void* bg2, int B, int n, int k, int d, float scale, int save, void* T0,
                        void* delta, void* U, void* T1, void* Lg, float* mixed, float* mx, float* sm, void* stream);
int64_t sug_ptran_colsum_workspace(int64_t rows);
int sug_ptran_relu_bwd_db(void* G, const void* T1, int64_t rows, int d, int dtype, float* db, float* ws, void* stream);

/* out[i] = (float) sum over g of red[g][i], i < n (fp64 partial rows of `groups` domain groups, in order). */
int sug_fold_groups(const double* red, int groups, int n, float* out, void* stream);

/* ---- Gaussian multi-kernel MMD --------------------------------------------------
 * replaces _mix_rbf_kernel + _mmd2(biased=True), model/mmd.py:239-254, :274-312.
 * Z = [X;Y] : [2m, D] rows (ld = ldz).  e_ij = n_i - 2<z_i,z_j> + n_j with n the
 * Gram diagonal; K = sum_s exp(-e/(2 sigma_s^2)); w (NULL or [m]) multiplies the
 * column sums of K_XY.  sums[0..2] += S_XX, S_YY, S_wXY (fp64, caller zeroes);
 * mmd2 = (S_XX + S_YY - 2 S_wXY)/m^2 is formed by the caller.
 * wt (NULL or [2m,2m]) receives c'_ij * dK_ij/de_ij for the backward
 * (dZ = 2*(diag(rowsum(wt)) - wt) . Z).  neg_gamma: device array [nsigma] holding
 * -1/(2 sigma_s^2) rounded to fp32 (the way torch rounds the python scalar,
 * model/mmd.py:251-252); nsigma <= 8. */
int sug_mmd_rbf(const float* z, int64_t ldz, int m, int D, const float* w,
                const float* neg_gamma, int nsigma, double* sums, float* wt, void* stream);

/* sug_mmd_rbf with the scalar formed on the device: zeroes sums (3 doubles of scratch), runs the kernel and
 * writes value[0] = (S_XX + S_YY - 2 S_wXY) / m^2 as fp32 (model/mmd.py:300-312, biased estimator). */
int sug_mmd_rbf_value(const float* z, int64_t ldz, int m, int D, const float* w, const float* neg_gamma,
                      int nsigma, double* sums, float* wt, float* value, void* stream);

/* Backward of sug_mmd_rbf: dz[i,:] = gscale[0] * 2 * (rowsum(wt)[i]*z[i,:] - (wt.z)[i,:])
 * (the autograd of model/mmd.py:239-312 w.r.t. the features); gscale: device scalar = dL/dmmd2. */
int sug_mmd_rbf_bwd(const float* z, int64_t ldz, const float* wt, int m, int D, const float* gscale,
                    float* dz, int64_t lddz, void* stream);

/* Row-block forms for the batch-sharded global MMD (no reference counterpart: the WIP DDP
 * trainer uses per-rank MMD, train_dg.py:357-368).  z is the gathered [2m, D] matrix (all ranks' X rows,
 * then all ranks' Y rows); this rank owns X rows [row0, row0+mloc) and Y rows m + [row0, row0+mloc).
 * sug_mmd_rbf_rows adds the contributions of the pairs (i in own rows, j in all columns) to sums[0..2]
 * (the ranks' partial sums are then added: an all-reduce of 3 doubles) and writes wt for the own rows
 * only, [2*mloc, 2m]; sug_mmd_rbf_rows_bwd turns that into the gradient of the OWN rows,
 * dz [2*mloc, D] = gmul * gscale[0] * 2 * (rowsum(wt) z_i - wt.z): no collective in the backward.
 * row0 = 0, mloc = m are sug_mmd_rbf / sug_mmd_rbf_bwd. */
int sug_mmd_rbf_rows(const float* z, int64_t ldz, int m, int D, const float* w, const float* neg_gamma,
                     int nsigma, int row0, int mloc, double* sums, float* wt, void* stream);
int sug_mmd_rbf_rows_bwd(const float* z, int64_t ldz, const float* wt, int m, int D, int row0, int mloc,
                         const float* gscale, float gmul, float* dz, int64_t lddz, void* stream);

/* SDA sample weights from class probabilities: prob_weights_soft + distance2weights,
 * model/mmd.py:134-148 and :178-202 (the reference computes them on the CPU with scipy's kl_div).
 * pred_* [m,10] logits (row strides lds/ldt), label_* int64 [m]; method 0 "none", 1 "naive_inverse",
 * 2 "exp_inverse", 3 "mean2one"; weights [m]. */
int sug_sda_prob_weights(const float* pred_s, int64_t lds, const float* pred_t, int64_t ldt,
                         const int64_t* label_s, const int64_t* label_t, int m, int num_class,
                         float label_weight, int method, float* weights, void* stream);
/* The same for the logits of n <= 4 heads on one batch (same labels), one launch: HOST arrays of n device pointers / strides. */
int sug_sda_prob_weights_multi(int n, const void* const* pred_s, const int64_t* lds, const void* const* pred_t,
                               const int64_t* ldt, const int64_t* label_s, const int64_t* label_t, int m, int num_class,
                               float label_weight, int method, void* const* weights, void* stream);

/* Z [2m, D+num_class] = [feat_s ; feat_t | one-hot(label) * label_scale]: the label-augmented operand of soft_mmd
 * (model/mmd.py:56-66, create_one_hot_labels utils/common_utils.py:161-164) in one launch; feat_* [m,D] with row
 * strides lds / ldt, label_* int64 [m]. */
int sug_mmd_assemble(const float* feat_s, int64_t lds, const float* feat_t, int64_t ldt, const int64_t* label_s,
                     const int64_t* label_t, int m, int D, int num_class, float label_scale, float* z, void* stream);

/* Up to 4 soft-MMD terms of ONE batch (same m samples per domain, same labels; a SUG step has three:
 * train_dg_single_gpu.py:300-322 -> mmd_cal -> soft_mmd, model/mmd.py:25-41, :56-66) with one launch per stage instead of
 * one per stage and term: assemble (sug_mmd_assemble; it also clears sums), kernel sums + derivative weights
 * (sug_mmd_rbf: each term through the kernel that call would choose), the n values; backward: sug_mmd_rbf_bwd of every
 * term in one launch.  Per-term results are bit-identical to the single-term calls.  HOST arrays of n entries:
 * feat_s / feat_t [m, D[i]] (row strides lds / ldt), label_scale, w (device [m] or null), z (device [2m, D[i] + num_class],
 * written), wt (device [2m, 2m] written, or null: no backward).  sums: device double [3n] scratch; values: device float [n].
 * Backward: gscale[i] = device scalar (upstream gradient of value i) or null (term skipped), dz[i] device [2m, D[i]]
 * dense = the gradient of [feat_s ; feat_t] (the label columns of z are constants), written entirely. */
int sug_soft_mmd_multi_fwd(int n, const void* const* feat_s, const int64_t* lds, const void* const* feat_t,
                           const int64_t* ldt, const int32_t* D, const float* label_scale, const int64_t* label_s,
                           const int64_t* label_t, int m, int num_class, const void* const* w, const float* neg_gamma,
                           int nsigma, void* const* z, void* const* wt, double* sums, float* values, void* stream);
int sug_soft_mmd_multi_bwd(int n, const void* const* z, const int32_t* D, const void* const* wt,
                           const void* const* gscale, int m, int num_class, void* const* dz, void* stream);

/* The EdgeConv GEMM operand of a conv_2d weight W [Co, 2C] (get_graph_feature's cat(x_j - x_i, x_i) folded into
 * the weights, model/model_utils.py:188-210): backward = 0: in = W, out [2Co, C] = [W[:, :C] ; W[:, C:] - W[:, :C]];
 * backward = 1: in = d out [2Co, C], out = dW [Co, 2C]. */
int sug_edge_weight_split(const float* in, int Co, int C, int backward, float* out, void* stream);
/* The same for n <= 8 weights in ONE launch (the four EdgeConv layers of Model.py:54-121): HOST arrays of n device
 * pointers and shapes; an entry whose in or out pointer is null is skipped (a layer without a gradient). */
int sug_edge_weight_split_multi(const void* const* in_host, const int32_t* Co_host, const int32_t* C_host, int n,
                                int backward, void* const* out_host, void* stream);

/* Chamfer distance per cloud pair (SDA geometric weights; geometric_weights(),
 * model/mmd.py:107-131 -- third-party op in the reference, parity unpinned):
 * out[b] = mean_i min_j |a_i-b_j|^2 + mean_j min_i |a_i-b_j|^2, direct-form distance.
 * a [B,N,3], b [B,M,3], out [B] (written, not accumulated); ws: sug_chamfer_workspace(B, N, M) floats -- the partial sums
 * of the workgroups, folded per cloud in block order (no float atomics: the value is reproducible run to run). */
int64_t sug_chamfer_workspace(int B, int N, int M);
int sug_chamfer(const float* a, const float* b, int B, int N, int M, float* out, float* ws, void* stream);
/* sug_chamfer followed by distance2weights (model/mmd.py:178-202) in the fold's own launch: out[b] = the SDA geometric weight
 * of pair b; method 1 naive_inverse, 2 exp_inverse, 3 mean2one (1 / mean truncated to an integer, :200).  Sums over the
 * batch in fp64, fixed order. */
int sug_chamfer_weights(const float* a, const float* b, int B, int N, int M, int method, float* out, float* ws, void* stream);

/* ---- the scalar tail of a step ---------------------------------------------------------------------------
 * Cross entropy of both classifier heads on the source rows of the paired logits (train_dg_single_gpu.py:269-292 with
 * nn.CrossEntropyLoss(), :167): loss[0] = w * (CE(logits1[:M], label) + CE(logits2[:M], label)), CE = mean over the M
 * rows of -log_softmax(row)[label]; w = 0.5 * SRC_LOSS_WEIGHT * CLS_WEIGHT folded by the caller.  logits* [>= M, C] with
 * row stride ld, label int64 [M], lse [2M + 1] (saved log-sum-exp of the rows; lse[2M] = number of rows that count).
 * 2M <= 512, C <= 32.  Labels as nn.CrossEntropyLoss takes them: a row whose label == ignore_index (-100 by default in
 * torch) contributes nothing and is left out of the mean; any other label outside [0, C) -- where torch raises -- makes
 * the loss (and the gradient) NaN: never scored as some class. */
int sug_ce_pair_fwd(const float* logits1, const float* logits2, int64_t ld, const int64_t* label, int M, int C, float w,
                    int64_t ignore_index, float* loss, float* lse, void* stream);
/* Its gradient for the WHOLE paired logits: d1, d2 [Mtot, C] = g[0] * w * (softmax - onehot) / lse[2M] in the counting rows
 * < M, zero in ignored rows and in rows M .. Mtot-1 (the target half of a paired batch: no torch.stack / zero fill rebuilds
 * the pair's gradient). */
int sug_ce_pair_bwd(const float* logits1, const float* logits2, int64_t ld, const int64_t* label, int M, int Mtot, int C,
                    float w, int64_t ignore_index, const float* g, const float* lse, float* d1, float* d2, void* stream);
/* out3 = { loss_cls + wg*v_geo + ws*(v_sem1 + v_sem2), wg*v_geo, ws*(v_sem1 + v_sem2) } (train_dg_single_gpu.py:314-324; the
 * weights MMD_WEIGHT * GEO_SCALE and 0.5 * MMD_WEIGHT * SEM_SCALE folded by the caller); null v_* = term absent.
 * Backward: out4 = g[0] * {1, wg, ws, ws}. */
int sug_loss_combine_fwd(const float* loss_cls, const float* v_geo, const float* v_sem1, const float* v_sem2, float wg, float ws,
                         float* out3, void* stream);
int sug_loss_combine_bwd(const float* g, float wg, float ws, float* out4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUG_AMD_H */
