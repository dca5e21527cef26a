"""Diagnostic: which Python call sites make real copies (contiguous() / reshape() of strided tensors,
.to()/.float() conversions) during one training step."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
tr = SUGStep(Net_MDA('DGCNN').to(dev).train())
data = synth(32, 1024, 666, dev)
for _ in range(2):
    tr.step(*data)
log = collections.Counter()
orig_c, orig_r = torch.Tensor.contiguous, torch.Tensor.reshape
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if '/repo/sug_amd' in f.filename or '/repo/bench' in f.filename:
            return '%s:%d' % (f.filename.split('/repo/')[-1], f.lineno)
    return '?'
def contiguous(self, *a, **k):
    if not self.is_contiguous():
        log[('contiguous', site(), tuple(self.shape))] += 1
    return orig_c(self, *a, **k)
def reshape(self, *shape):
    out = orig_r(self, *shape)
    if out.data_ptr() != self.data_ptr() or (out.numel() and out._base is None and self._base is None and out is not self and not self.is_contiguous()):
        if not self.is_contiguous() and out.data_ptr() != self.data_ptr():
            log[('reshape-copy', site(), tuple(self.shape))] += 1
    return out
torch.Tensor.contiguous, torch.Tensor.reshape = contiguous, reshape
tr.step(*data)
torch.cuda.synchronize()
torch.Tensor.contiguous, torch.Tensor.reshape = orig_c, orig_r
for (kind, s, shp), n in sorted(log.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][2]).numel()):
    print('%3d x %-14s %-44s %s  (%.1f MB)' % (n, kind, s, shp, torch.Size(shp).numel() * 4 / 1e6))
