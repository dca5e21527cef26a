"""Host-side logic that needs no GPU: the model boundary's GPU-only contract, the capturable Adam's bucket plan."""
import pytest
import torch


def test_model_boundary_rejects_cpu_input_with_a_clear_message():
    """There is no CPU / eager fallback behind Net_MDA (the product path must fail loudly)."""
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA('DGCNN')
    with pytest.raises(RuntimeError, match='HIP device'):
        net(torch.zeros(2, 3, 64, 1), semantic_adaption=True)
    with pytest.raises(RuntimeError, match='HIP device'):
        net.forward_pair(torch.zeros(4, 3, 64, 1))


def test_ops_reject_cpu_tensors():
    from sug_amd import ops
    with pytest.raises(RuntimeError, match='HIP device only'):
        ops.knn(torch.zeros(1, 32, 3), 4)
    with pytest.raises(RuntimeError, match='HIP device only'):
        ops.linear_rows(torch.zeros(8, 4), torch.zeros(4, 4))


def test_graph_key_excludes_learning_rates():
    """SUGStep's graph key: by-value hyper-parameters of every param group of the three optimizers, not their lr
    (sug_amd.optim.Adam keeps lr on the device); torch optimizers contribute their lr too."""
    from sug_amd.optim import Adam
    w = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    opt = Adam([{'params': [w[0]]}, {'params': [w[1]], 'weight_decay': 0.1}, {'params': [w[2]]}], lr=1e-3, graph_capturable=True)
    k0 = opt.graph_key()
    opt.param_groups[1]['lr'] = 5e-4
    assert opt.graph_key() == k0
    opt.param_groups[1]['weight_decay'] = 0.2
    assert opt.graph_key() != k0
    assert len(k0) == 3


def test_sharded_steps_weight_a_term_the_way_mmd_cal_does():
    """SUGStep's batch-sharded / segmented steps computed the semantic SDA weights with prob_weights_soft whenever
    SEM_WEIGHTS was set and never looked at ENTROPY_WEIGHTS, which cal_sample_weights ranks first (model/mmd.py:44-53).  They
    now go through mmd.sda_weights_of: mmd_cal's own rules.  (Pure torch on probability inputs: runs without a GPU.)"""
    import inspect
    from sug_amd import train_step
    from sug_amd.model import mmd
    g = torch.Generator().manual_seed(0)
    ps, pt = torch.softmax(torch.randn(6, 10, generator=g), 1), torch.softmax(torch.randn(6, 10, generator=g), 1)
    ls, lt = torch.randint(0, 10, (6,), generator=g), torch.randint(0, 10, (6,), generator=g)
    both = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'ENTROPY_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5}
    w = mmd.sda_weights_of(both, ps, pt, ls, lt)
    assert torch.equal(w, mmd.entropy_weights(ps, pt, 'none'))                 # ENTROPY ahead of SEM
    assert mmd.sda_weights_of({'NAME': 'SOFT_MMD', 'ENTROPY_WEIGHTS': 'none'}, ps, pt, ls, lt) is None      # :28: GEO / SEM switch weights on
    assert mmd.sda_weights_of({'NAME': 'SOFT_MMD'}, ps, pt, ls, lt) is None
    src = inspect.getsource(train_step.SUGStep)
    assert src.count('mmd.sda_weights_of(') == 2 and 'mmd.prob_weights_soft(' not in src


def test_call_graph_manager_is_not_copied_or_pickled_with_the_model():
    import copy
    import pickle
    from sug_amd import call_graphs
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA('Pointnet')
    mgr = call_graphs.manager_for(net)
    assert mgr is not None and call_graphs.manager_for(net) is mgr
    twin = copy.deepcopy(net)
    assert twin.__dict__.get('_call_graph_mgr') is None
    assert call_graphs.manager_for(twin) is not mgr
    assert pickle.loads(pickle.dumps(mgr)) is None
