"""Diagnostic: gradient error of PointNet++'s set-abstraction layers (the real modules with the test's parameter fill) against
an fp64 torch restatement fed with the SAME inputs and output gradient.  usage: python tests/diagnostics/diag_sa3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import ref_cpu as O          # (diagnostic only: parameter fill of the tests)
from sug_amd.model.Model import Net_MDA
torch.manual_seed(0)
rel = lambda a, b: float((a.double().cpu() - b.cpu()).norm() / (b.cpu().norm() + 1e-300))
net = Net_MDA('Pointnet2')
net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 5))
net = net.cuda().train()
sa = net.g.sa3
B, n = 4, 128
xyz = (torch.rand(B, n, 3) - 0.5).cuda()
pts = torch.relu(torch.randn(B, n, 256) * 0.6 + 0.2).cuda().requires_grad_(True)
if len(sys.argv) > 1 and sys.argv[1] == 'real':      # the inputs sa3 sees inside the network on the test's clouds
    seen = []
    orig = sa.rows
    sa.rows = lambda a, b, *k, **kw: (seen.append((a.detach().clone(), b.detach().clone())), orig(a, b, *k, **kw))[1]
    g0 = torch.Generator().manual_seed(3)
    clouds = O.synth_clouds(4, 2048, g0)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    torch.manual_seed(6)
    outs = []
    sa.rows = lambda a, b, *k, **kw: (seen.append((a.detach().clone(), b.detach().clone())), outs.append(orig(a, b, *k, **kw)), outs[-1])[2]
    y1, y2, s1, s2 = net(clouds.cuda(), semantic_adaption=True)
    gcap = []
    outs[0][1].register_hook(lambda g: gcap.append(g.detach().clone()))
    w1, w2 = torch.randn(4, 10, generator=g0).cuda(), torch.randn(4, 256, generator=g0).cuda()
    ((y1 * w1).sum() + (y2 * w1).sum() + (s1 * w2).sum() + (s2 * w2).sum()).backward()
    innet = {k: v.grad.detach().clone() for k, v in sa.named_parameters() if v.grad is not None}
    net.zero_grad()
    sa.rows = orig
    xyz, pts = seen[0][0], seen[0][1].requires_grad_(True)
    PROBE = gcap[0]
    z = torch.cat((xyz, pts.detach()), -1).reshape(B * n, -1)
    print('real inputs: feature mean %.3f std %.3f max %.3f; fraction of exact zeros %.3f; duplicate rows %d' % (
        float(pts.mean()), float(pts.std()), float(pts.max()), float((pts == 0).float().mean()),
        B * n - len(torch.unique(z, dim=0))))
_, out = sa.rows(xyz, pts)
probe = PROBE.view_as(out) if 'PROBE' in globals() else torch.randn_like(out)
params = [p for p in sa.parameters()]
got = torch.autograd.grad((out * probe).sum(), [pts] + params, allow_unused=True)
# fp64 restatement of sample_and_group_all + the three layers + max
xd, pd = xyz.double().cpu(), pts.detach().double().cpu().requires_grad_(True)
P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in sa.named_parameters()}
g = torch.cat((xd, pd), -1).reshape(B * n, -1)
for i in range(3):
    w = P['mlp_convs.%d.weight' % i].view(P['mlp_convs.%d.weight' % i].shape[0], -1)
    g = torch.relu(F.batch_norm(g @ w.t() + P['mlp_convs.%d.bias' % i], None, None, P['mlp_bns.%d.weight' % i],
                                P['mlp_bns.%d.bias' % i], True, 0.1, 1e-5))
outd = g.view(B, n, -1).max(1)[0]
names = [k for k, _ in sa.named_parameters()]
ref = torch.autograd.grad((outd * probe.double().cpu().view(B, -1)).sum(), [pd] + [P[k] for k in names], allow_unused=True)
print('forward out rel err %.2e' % rel(out.view(B, -1), outd))
for nm, a, b in zip(['d points'] + names, got, ref):
    if a is not None and b is not None and float(b.norm()) > 1e-9:
        print('%-24s rel err %.2e' % (nm, rel(a.view(b.shape), b)))

if 'innet' in globals():
    print('in-network HIP gradients of sa3 against the fp64 restatement on the same inputs / output gradient:')
    for nm, b in zip(names, ref[1:]):
        if nm in innet and b is not None and float(b.norm()) > 1e-9:
            print('%-24s rel err %.2e' % (nm, rel(innet[nm].view(b.shape), b)))

if 'innet' in globals():
    # the same forward in the fp64 oracle: how far are the inputs of sa3 (the outputs of sa2) and the pooled feature apart?
    p64 = O.as_params({k: (v.double() if v.dtype.is_floating_point else v) for k, v in
                       O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 5).items()})
    torch.manual_seed(6)
    xx = clouds.double().squeeze(-1)
    with torch.no_grad():
        l1_xyz, l1_pts, node = O.set_abstraction(p64, 'g.sa1.', xx, None, 512, 0.2, 32, adapt=True, training=True)
        l2_xyz, l2_pts = O.set_abstraction(p64, 'g.sa2.', l1_xyz, l1_pts, 128, 0.4, 64, training=True)
        _, l3 = O.set_abstraction(p64, 'g.sa3.', l2_xyz, l2_pts, None, None, None, group_all=True, training=True)
    hip_l2 = seen[0][1].detach().double().cpu()                 # [B,128,256]
    ora_l2 = l2_pts.permute(0, 2, 1)
    d = (hip_l2 - ora_l2).abs()
    print('sa2 output, HIP vs fp64 oracle: rel L2 %.2e, max abs %.2e (max |value| %.2f); entries off by > 1e-3: %d of %d' % (
        float((hip_l2 - ora_l2).norm() / ora_l2.norm()), float(d.max()), float(ora_l2.abs().max()), int((d > 1e-3).sum()), d.numel()))
    print('pooled feature, HIP vs fp64 oracle: rel L2 %.2e' % float((outs[0][1].detach().double().cpu().view(4, -1) - l3.view(4, -1)).norm() / l3.norm()))
    # conditioning: the fp64 restatement of sa3 on the HIP inputs against the same restatement on the fp64 oracle's inputs
    def restate(xyz_in, pts_in):
        Pq = {k: v.detach().double().cpu().requires_grad_(True) for k, v in sa.named_parameters()}
        gg = torch.cat((xyz_in, pts_in), -1).reshape(B * n, -1)
        pre = []
        for i in range(3):
            w = Pq['mlp_convs.%d.weight' % i].view(Pq['mlp_convs.%d.weight' % i].shape[0], -1)
            zz = gg @ w.t() + Pq['mlp_convs.%d.bias' % i]
            pre.append(zz.detach())
            gg = torch.relu(F.batch_norm(zz, None, None, Pq['mlp_bns.%d.weight' % i], Pq['mlp_bns.%d.bias' % i], True, 0.1, 1e-5))
        oo = gg.view(B, n, -1).max(1)[0]
        gr = torch.autograd.grad((oo * PROBE.double().cpu().view(B, -1)).sum(), [Pq[k] for k in names], allow_unused=True)
        return gr, pre
    ga, prea = restate(xyz.double().cpu(), seen[0][1].detach().double().cpu())
    gb, preb = restate(l2_xyz.permute(0, 2, 1), ora_l2)
    print('fp64 restatement, HIP inputs vs oracle inputs (same output gradient):')
    for nm, a_, b_ in zip(names, ga, gb):
        if a_ is not None and float(b_.norm()) > 1e-9:
            print('   %-22s rel diff %.2e' % (nm, float((a_ - b_).norm() / b_.norm())))
    for i in range(3):
        sd = prea[i].std(0)
        print('   layer %d pre-BN: smallest channel std %.2e (median %.2e); inputs rel diff %.2e' % (
            i, float(sd.min()), float(sd.median()), float((prea[i] - preb[i]).norm() / preb[i].norm())))
