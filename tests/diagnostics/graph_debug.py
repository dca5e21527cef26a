"""Diagnostic: exercise SUGStep graph mode under different settings (env: B, STEPS, SHARE, FEED_EAGER)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd import ops
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep, _StartFeeder

B = int(os.environ.get('B', 32)); STEPS = int(os.environ.get('STEPS', 6))
share = os.environ.get('SHARE', '1') == '1'
dev = torch.device('cuda')
torch.manual_seed(666)
net = Net_MDA('DGCNN').to(dev).train()
drop = os.environ.get('DROP', 'keep')
for name, m in list(net.named_modules()):
    if isinstance(m, torch.nn.Dropout2d):
        if drop == '0':
            m.p = 0.0
        elif drop == 'plain':
            parent = net.get_submodule(name.rsplit('.', 1)[0])
            setattr(parent, name.rsplit('.', 1)[1], torch.nn.Dropout(p=m.p))
data = synth(B, 1024, 666, dev)
if os.environ.get('FILL') == '1':
    from oracle import ref_cpu as O
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 666))
if os.environ.get('GOLD') == '1':
    from tests.conftest import load_golden
    G = load_golden('step_dgcnn.npz')
    data = [G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda()]
if os.environ.get('FEED_EAGER') == '1':
    tr = SUGStep(net, share_prefix=share, use_graph=False)
    fd = _StartFeeder(dev)
    ops.START_PROVIDER = fd.record
    tr.step(*data); torch.cuda.synchronize()
    fd.build()
    for i in range(STEPS):
        fd.cursor = 0; fd.refill(); ops.START_PROVIDER = fd.provide
        out = tr.step(*data); torch.cuda.synchronize()
        print('feed-eager step', i, [float(x) for x in out], flush=True)
else:
    if os.environ.get('PRE_EAGER') == '1':
        net0 = Net_MDA('DGCNN').to(dev).train()
        tr0 = SUGStep(net0, share_prefix=share, use_graph=False)
        for i in range(3):
            tr0.step(*data)
        torch.cuda.synchronize()
        print('pre-eager trainer done', flush=True)
        if os.environ.get('KEEP') != '1':
            del tr0, net0
    tr = SUGStep(net, share_prefix=share, use_graph=True)
    for i in range(STEPS):
        out = tr.step(*data); torch.cuda.synchronize()
        print('graph step', i, [float(x) for x in out], flush=True)
        mode = os.environ.get('CHECK', '0')
        if mode == 'alloc':
            z = torch.zeros(1 << 20, device=dev); z += 1; del z
            continue
        if mode == 'sleep':
            import time; time.sleep(0.1)
            continue
        if mode == 'maxonly':
            gmax = max(float(v.abs().max()) for v in net.parameters())
            continue
        if mode == 'optstate':
            _ = [float(tr.optimizer_g.state[p]['step']) for p in list(tr.optimizer_g.state)[:1]]
            continue
        if mode != '1':
            continue
        bad = [k for k, v in net.state_dict().items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
        gmax = max(float(v.abs().max()) for v in net.parameters())
        print('   non-finite tensors:', bad[:5], 'max |param| %.3g' % gmax,
              'adam step', [float(tr.optimizer_g.state[p]['step']) for p in list(tr.optimizer_g.state)[:1]], flush=True)
if os.environ.get('TIME') == '1':
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        out = tr.step(*data)
    torch.cuda.synchronize(); print('ms/step %.2f' % ((time.perf_counter() - t0) / 40 * 1e3), [float(x) for x in out])
print('done')
