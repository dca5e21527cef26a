"""Regenerate sug_amd/tuning/{tunableop_gfx950.csv, dw_choice_gfx950.json} (needs an MI355X):
 1. run C2 training steps with TunableOp tuning on and every weight gradient routed to the library,
    so that all GEMM shapes of the step (forward, dx, dW) get a tuned entry;
 2. time sug_linear_dw against the tuned library GEMM for every weight-gradient shape of the step and
    list the shapes where the library wins by more than 5%.
usage: python tools/tune_gemms.py [outdir]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
outdir = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tuning'
os.makedirs(outdir, exist_ok=True)
csv = os.path.join(outdir, 'tunableop_gfx950.csv')
os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='1', PYTORCH_TUNABLEOP_FILENAME=csv)
import torch
import torch.cuda.tunable as tn
from bench import synth
from sug_amd import ops
from sug_amd._lib import lib
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
tn.set_filename(csv, insert_device_ordinal=False)
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5)
data = synth(32, 1024, 666, dev)
ops.DW_FORCE_LIBRARY, ops.DW_SHAPE_LOG = True, []
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
shapes = sorted(set(ops.DW_SHAPE_LOG))
ops.DW_FORCE_LIBRARY, ops.DW_SHAPE_LOG = False, None


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


library, rows = [], []
for R, M, N in shapes:
    g, x = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
    t_lib = timed(lambda: g.t() @ x)
    if M * N > 512 * 512:
        continue
    dw = torch.empty(M, N, device=dev)
    ws = torch.empty(int(lib().sug_linear_dw_workspace(R, M, N)), device=dev)
    st = ops._st()
    t_own = timed(lambda: lib().sug_linear_dw(g.data_ptr(), M, x.data_ptr(), N, R, M, N, dw.data_ptr(), ws.data_ptr(), st))
    rows.append((R, M, N, t_lib, t_own))
    if t_lib < 0.95 * t_own:
        library.append([R, M, N])
    print('dW rows=%d %dx%d: tuned library %.1f us, sug_linear_dw %.1f us -> %s' % (R, M, N, t_lib, t_own, 'library' if t_lib < 0.95 * t_own else 'own'))
if hasattr(tn, 'write_file'):
    tn.write_file(csv)
json.dump({'_comment': 'weight-gradient shapes (rows, M, N) where the TunableOp-tuned library GEMM beat sug_linear_dw by > 5% '
                       '(tools/tune_gemms.py, us: ' + '; '.join('%dx%d@%d lib %.1f own %.1f' % (m, n, r, a, b) for r, m, n, a, b in rows) + ')',
           'library': library}, open(os.path.join(outdir, 'dw_choice_gfx950.json'), 'w'), indent=1)
print('tuned %d GEMM shapes -> %s; %d of %d weight-gradient shapes go to the library' % (len(tn.get_results()), csv, len(library), len(rows)))
