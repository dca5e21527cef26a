"""Watch a long run of training steps for non-finite or absurd loss values / parameters (every step is checked, anomalies and
every 100th step are printed).  usage: python tools/graph_watch.py MODEL graph|eager STEPS [tuned] [FROM_STEP]
Found the memset-node problem of replayed step graphs: eager runs stay clean, graph runs showed constant
garbage MMD values after ~200 replays and NaN weights with library split-K weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
model = sys.argv[1]; use_graph = sys.argv[2] == 'graph'; steps = int(sys.argv[3])
if len(sys.argv) > 4 and sys.argv[4] == 'tuned': enable_tuned_gemms()
dev = torch.device('cuda')
torch.manual_seed(666)
net = Net_MDA(model).to(dev).train()
tr = SUGStep(net, use_graph=use_graph)
data = synth(32, 1024, 666, dev)
for i in range(steps):
    out = tr.step(*data)
    if i % 20 == 0 or i == steps - 1 or (len(sys.argv) > 5 and i >= int(sys.argv[5])):
        vals = [float(v) for v in out if v is not None]
        pmax = max(float(p.detach().abs().max()) for p in net.parameters())
        if i % 100 == 0 or any((v != v) or abs(v) > 1e3 for v in vals): print(i, vals, 'max|param| %.3g' % pmax, flush=True)
        if any(v != v for v in vals) and len(sys.argv) <= 5:
            bad = [k for k, p in net.named_parameters() if not torch.isfinite(p).all()]
            print('non-finite params:', bad[:10]); break
