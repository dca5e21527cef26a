"""Algorithmic work of one SUG training step ("Train step roofline = sum over the encoder passes + heads + 3 MMD
+ backward + optimizer") -- measurement support for bench.py, not product code.

`step_work(model, B, N)` returns FLOPs and compulsory HBM bytes of a step over B clouds per domain (2B per step) in the
formulation the kernels implement, NOT the reference's k-expanded one:

  * EdgeConv layer: one [N,C] x [C,2Co] product per cloud (`W.[x_j - x_i ; x_i] = W1.x_j + (W2 - W1).x_i`) + 6 VALU
    operations per gathered value (N.k.Co of them); the reference's formulation has k times the GEMM FLOPs;
  * kNN: N^2 (2C + 3) per cloud;
  * PointNet++ first SA layer: per point, not per (centroid, neighbour);
  * backward of a dense layer = 2 x its forward FLOPs (input gradient + weight gradient); kNN / FPS / ball query have none.

Passes of the exact two-pass step: the semantic pass runs forward + backward on 2B clouds; the node pass re-runs, forward
only, whatever sits behind the shared prefix -- that is what the reference's second encoder evaluation
costs once identical work is not repeated.  `single_pass=True` drops the node pass.
Bytes: every layer's compulsory traffic (inputs once, outputs once, SURVEY 8d's per-layer formulas), x3 for a layer with a
backward (forward; backward reads its inputs and the upstream gradient and writes one gradient), + 28 B per parameter
element and optimizer that owns it (Adam: read p, g, m, v; write p, m, v), + the weights themselves once per pass."""

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA rate
FP16_PEAK_TFLOPS = 2500.0       # dense fp16 MFMA
HBM_PEAK_GBS = 8000.0


class _Acc:
    def __init__(self):
        self.flops = 0.0        # fp32 arithmetic
        self.flops16 = 0.0      # arithmetic on the fp16 MFMA path (Point Transformer, --fp16)
        self.bytes = 0.0
        self.items = {}

    def add(self, name, flops=0.0, bytes_=0.0, f16=False):
        if f16:
            self.flops16 += flops
        else:
            self.flops += flops
        self.bytes += bytes_
        it = self.items.setdefault(name, [0.0, 0.0])
        it[0] += flops
        it[1] += bytes_


def _dense(acc, name, rows, cin, cout, train, e_in=4, e_out=4, f16=False):
    """rows x [cin -> cout] layer: forward (+ backward = input gradient + weight gradient when `train`)."""
    f = 2.0 * rows * cin * cout
    b = rows * (cin * e_in + cout * e_out) + 4.0 * cin * cout
    acc.add(name, f * (3 if train else 1), b * (3 if train else 1), f16)


def _knn(acc, clouds, N, C, k):
    acc.add('knn_C%d' % C, clouds * N * N * (2.0 * C + 3), clouds * (4.0 * C * N + 4.0 * N * k))


def _edgeconv(acc, clouds, N, C, Co, k, train):
    # decomposed GEMM + per-edge gather / max / BN work; SURVEY 8d fused-layer bytes 4N(C + Co) + 4Nk
    f = clouds * (2.0 * N * C * 2 * Co + 6.0 * N * k * Co)
    b = clouds * (4.0 * N * (C + Co) + 4.0 * N * k)
    acc.add('edgeconv_%dto%d' % (C, Co), f * (3 if train else 1), b * (3 if train else 1))


def _adapt(acc, clouds, N, train):
    """SA-node module (model_utils.py:92-128): FPS(64), ball query, offset prediction, 64-NN, residual conv, 3-NN interp."""
    acc.add('fps_64', clouds * 8.0 * N * 64, clouds * (12.0 * N + 4 * 64))
    acc.add('ball_query_64', clouds * 8.0 * N * 64, clouds * (12.0 * N + 12 * 64 + 4 * 64 * 64))
    acc.add('knn_query_64', clouds * 8.0 * N * 64, clouds * (12.0 * N + 12 * 64 + 4 * 64 * 64))
    _dense(acc, 'adapt.pred_offset', clouds * N, 64, 3, train)
    _dense(acc, 'adapt.residual', clouds * N, 64, 64, train)
    acc.add('adapt.group+interp', clouds * N * 64 * 8.0, clouds * N * (64 + 128 + 3 * 2) * 4.0 * (3 if train else 1))


def _heads(acc, M, feat_dim, ptran):
    """Two Pointnet_c heads on M pooled features + two CALayers on M/2 node features each (4096 -> 512 -> 4096)."""
    for _ in range(2):
        if not ptran:
            _dense(acc, 'heads', M, feat_dim, 512, True)
        _dense(acc, 'heads', M, 512, 256, True)
        _dense(acc, 'heads', M, 256, 10, True)


def _calayers(acc, M):
    for _ in range(2):
        _dense(acc, 'calayer', M // 2, 4096, 512, True)
        _dense(acc, 'calayer', M // 2, 512, 4096, True)


def _mmd(acc, m):
    for D in (4096 + 10, 256 + 10, 256 + 10):
        acc.add('mmd', 3 * 2.0 * (2 * m) ** 2 * D, 3 * 4.0 * 2 * m * D)


def _dgcnn(acc, clouds, N, k, node_pass):
    tr = not node_pass
    if not node_pass:                                   # prefix (kNN + conv1, kNN + conv2): shared by the node pass
        _knn(acc, clouds, N, 3, k)
        _edgeconv(acc, clouds, N, 3, 64, k, True)
        _knn(acc, clouds, N, 64, k)
        _edgeconv(acc, clouds, N, 64, 64, k, True)
    _adapt(acc, clouds, N, True)                        # (node pass: its outputs carry the geometric MMD's gradient)
    _dense(acc, 'conv1d', clouds * N, 128, 64, tr)
    _knn(acc, clouds, N, 64, k)
    _edgeconv(acc, clouds, N, 64, 128, k, tr)
    _knn(acc, clouds, N, 128, k)
    _edgeconv(acc, clouds, N, 128, 256, k, tr)
    _dense(acc, 'conv5', clouds * N, 512, 512, tr)
    acc.add('bn5+pool', clouds * N * 512 * 6.0, clouds * N * 512 * 4.0 * (4 if tr else 1))


def _tnet(acc, clouds, N, K, train=True):
    _dense(acc, 'tnet', clouds * N, K, 64, train)
    _dense(acc, 'tnet', clouds * N, 64, 128, train)
    acc.add('tnet.conv3+max', 2.0 * clouds * N * 128 * 1024 * (3 if train else 1), clouds * (4.0 * N * 128 + 8 * 1024) * (3 if train else 1))
    for a, b in ((1024, 512), (512, 256), (256, K * K)):
        _dense(acc, 'tnet.fc', clouds, a, b, train)


def _pointnet(acc, clouds, N, node_pass):
    tr = not node_pass
    if not node_pass:                                   # both T-Nets, conv1, conv2: shared prefix
        _tnet(acc, clouds, N, 3)
        _dense(acc, 'conv1', clouds * N, 3, 64, True)
        _dense(acc, 'conv2', clouds * N, 64, 64, True)
        _tnet(acc, clouds, N, 64)
    _adapt(acc, clouds, N, True)
    _dense(acc, 'conv4', clouds * N, 128, 128, tr)
    acc.add('conv5+max', 2.0 * clouds * N * 128 * 1024 * (3 if tr else 1), clouds * (4.0 * N * 128 + 8 * 1024) * (3 if tr else 1))


def _pointnet2(acc, clouds, N, node_pass):
    tr = not node_pass
    S1, n1, S2, n2 = 512, 32, 128, 64
    acc.add('fps', clouds * 8.0 * (N * S1 + S1 * S2), clouds * (12.0 * (N + S1) + 4 * (S1 + S2)))
    acc.add('ball_query', clouds * 8.0 * (N * S1 + S1 * S2), clouds * (12.0 * (N + S1) + 12 * (S1 + S2) + 4 * (S1 * n1 + S2 * n2)))
    r1, r2 = clouds * S1 * n1, clouds * S2 * n2
    _dense(acc, 'sa1.l0 (per point)', clouds * (N + S1), 3, 64, True)
    acc.add('sa1.l0 gather+bn', r1 * 64 * 4.0, r1 * 64 * 4.0 * 3)
    _dense(acc, 'sa1.l1', r1, 64, 64, True)
    acc.add('sa1.l2+max', 2.0 * r1 * 64 * 128 * (3 if tr else 1), (4.0 * r1 * 64 + 8.0 * clouds * S1 * 128) * (3 if tr else 1))
    _dense(acc, 'sa2.l0 (per point)', clouds * (S1 + S2), 131, 128, tr)
    acc.add('sa2.l0 gather+bn', r2 * 128 * 4.0, r2 * 128 * 4.0 * (3 if tr else 1))
    _dense(acc, 'sa2.l1', r2, 128, 128, tr)
    acc.add('sa2.l2+max', 2.0 * r2 * 128 * 256 * (3 if tr else 1), (4.0 * r2 * 128 + 8.0 * clouds * S2 * 256) * (3 if tr else 1))
    for a, b in ((259, 256), (256, 512), (512, 1024)):
        _dense(acc, 'sa3', clouds * S2, a, b, tr)


def _ptran(acc, clouds, N, node_pass, fp16):
    e = 2 if fp16 else 4
    stages = [(N, 32)] + [(1024 // 4 ** (i + 1), 32 * 2 ** (i + 1)) for i in range(4)]      # (points, d_points) per block
    for bi, (n, dp) in enumerate(stages):
        if node_pass and bi == 0:
            continue                                    # fc1 + transformer1: shared prefix
        tr = (not node_pass) or bi <= 2                 # node features come from block 2
        k = min(16, n)
        rows, rk = clouds * n, clouds * n * k
        if bi == 0:
            _dense(acc, 'fc1', rows, 3, 32, tr)
            _dense(acc, 'fc1', rows, 32, 32, tr)
        else:
            pn, pd = stages[bi - 1]
            acc.add('td.fps+knn', clouds * 8.0 * pn * n * 2, clouds * (12.0 * pn + 4.0 * n * k))
            _dense(acc, 'td.mlp', rk, pd + 3, dp, tr)
            _dense(acc, 'td.mlp', rk, dp, dp, tr)
        acc.add('block.knn', clouds * 8.0 * n * n, clouds * (12.0 * n + 4.0 * n * k))
        _dense(acc, 'block.fc1', rows, dp, 512, tr, f16=fp16)
        for _ in range(3):
            _dense(acc, 'block.qkv', rows, 512, 512, tr, e_out=e, f16=fp16)
        _dense(acc, 'block.delta0', rk, 3, 512, tr, e_out=e)
        for _ in range(3):                              # fc_delta[2], fc_gamma[0], fc_gamma[2] on the k-expanded rows
            _dense(acc, 'block.kexp512', rk, 512, 512, tr, e_in=e, e_out=e, f16=fp16)
        acc.add('block.attn', rk * 512 * 10.0 * (3 if tr else 1), rk * 512 * e * 2.0 * (3 if tr else 1))
        _dense(acc, 'block.fc2', rows, 512, dp, tr, f16=fp16)


PARAMS = {'DGCNN': None, 'Pointnet': None, 'Pointnet2': None, 'PTran': None}


def step_work(model, B, N, k=20, single_pass=False, fp16=False, n_params=None, adam_elems=None):
    """-> dict(flops, flops16, bytes, items) of one step over B clouds per domain.  `adam_elems`: parameter elements
    summed over the three optimizers (a parameter of g is updated by optimizer_g AND optimizer_dis)."""
    acc = _Acc()
    clouds = 2 * B
    enc = {'DGCNN': lambda np_: _dgcnn(acc, clouds, N, k, np_), 'Pointnet': lambda np_: _pointnet(acc, clouds, N, np_),
           'Pointnet2': lambda np_: _pointnet2(acc, clouds, N, np_), 'PTran': lambda np_: _ptran(acc, clouds, N, np_, fp16)}[model]
    enc(False)
    if not single_pass:
        enc(True)
    _heads(acc, clouds, 512 if model == 'PTran' else 1024, model == 'PTran')
    _calayers(acc, clouds)
    _mmd(acc, B)
    if adam_elems:
        acc.add('adam', 12.0 * adam_elems, 28.0 * adam_elems)
    if n_params:
        acc.add('weights', 0.0, 4.0 * n_params * (2 if single_pass else 3))
    return {'flops': acc.flops, 'flops16': acc.flops16, 'bytes': acc.bytes,
            'items': {k_: {'gflop': round(v[0] / 1e9, 3), 'mb': round(v[1] / 1e6, 2)} for k_, v in acc.items.items()}}


def step_roofline(model, B, N, ms, single_pass=False, fp16=False, n_params=None, adam_elems=None):
    """The `roofline.step` object: algorithmic FLOPs and bytes of the step / its measured time, against the fp32 MFMA peak
    (fp16 part against the fp16 peak: time-additive bound) and the HBM peak."""
    w = step_work(model, B, N, single_pass=single_pass, fp16=fp16, n_params=n_params, adam_elems=adam_elems)
    t = ms * 1e-3
    t_mfma = w['flops'] / (FP32_PEAK_TFLOPS * 1e12) + w['flops16'] / (FP16_PEAK_TFLOPS * 1e12)
    t_hbm = w['bytes'] / (HBM_PEAK_GBS * 1e9)
    top = sorted(w['items'].items(), key=lambda kv: -kv[1]['gflop'])[:6]
    return {'gflop': round((w['flops'] + w['flops16']) / 1e9, 2), 'gflop_fp16': round(w['flops16'] / 1e9, 2),
            'gbytes': round(w['bytes'] / 1e9, 4), 'ms': round(ms, 4),
            'achieved_tflops': round((w['flops'] + w['flops16']) / t / 1e12, 2),
            'achieved_gbps': round(w['bytes'] / t / 1e9, 1),
            'frac_mfma': round(t_mfma / t, 4), 'frac_hbm': round(t_hbm / t, 4),
            'bound': 'mfma' if t_mfma >= t_hbm else 'hbm',
            'peaks': {'fp32_tflops': FP32_PEAK_TFLOPS, 'fp16_tflops': FP16_PEAK_TFLOPS, 'hbm_gbps': HBM_PEAK_GBS},
            'largest_items_gflop': {k_: v['gflop'] for k_, v in top},
            'note': 'frac_mfma = (fp32 FLOPs / fp32 peak + fp16 FLOPs / fp16 peak) / step time; frac_hbm = compulsory bytes / HBM '
                    'peak / step time; work counted in the decomposed formulation the kernels implement (bench_work.py), the '
                    'node pass forward-only behind the shared prefix'}
