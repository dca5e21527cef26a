"""knn_pc_kernel forms (producer + consumer waves per workgroup: 4+4 = 256 queries, 2+2 and 4+2 = 128 queries), us per launch at the C2 shapes.
The form is chosen per process (SUG_KNN_PW is read once), so each form runs in a child process.  usage: python tools/bench_knn_pw.py [CLOUDS]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = '''
import sys, torch
sys.path.insert(0, %r)
from sug_amd import ops
torch.manual_seed(0)
for C in (3, 64, 128):
    x = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1024, C, device="cuda")
    for _ in range(3): ops.knn(x, 20)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): idx = ops.knn(x, 20)
    b.record(); torch.cuda.synchronize()
    print('C=%%3d %%7.1f us  checksum %%d' %% (C, a.elapsed_time(b) / 20 * 1e3, int(idx.long().sum())))
''' % ROOT
for pw in (os.environ.get('FORMS') or '44,22,42').split(','):
    print('SUG_KNN_PW=' + pw, flush=True)
    subprocess.run([sys.executable, '-c', CHILD] + sys.argv[1:2], env=dict(os.environ, SUG_KNN_PW=pw), check=True)
