"""Point Transformer: fp32 (parity) vs bf16 / fp16 GEMM mode -- step time and deviation of the logits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model import Ptran_transformer as PT
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
data = synth(B, 1024, 666, dev)
torch.manual_seed(0)
net = Net_MDA('PTran').to(dev).train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout2d):
        m.p = 0.0
ref = None
for name, dt in (('fp32', None), ('bf16', torch.bfloat16), ('fp16', torch.float16)):
    PT.GEMM_DTYPE = dt
    torch.manual_seed(1)
    with torch.no_grad():
        y1, y2, f1, f2 = net(data[0], semantic_adaption=True)
    if ref is None:
        ref = (y1, f1)
    dev_logit = float((y1 - ref[0]).abs().max() / ref[0].abs().max())
    dev_feat = float((f1 - ref[1]).abs().max() / ref[1].abs().max())
    tr = SUGStep(net, lr=0.0)
    for _ in range(2):
        tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(*data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print('%s: %.1f ms/step (%.0f clouds/s), logits max rel dev %.2e, sem feature max rel dev %.2e' % (name, ms, 2 * B / ms * 1e3, dev_logit, dev_feat))
