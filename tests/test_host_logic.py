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
