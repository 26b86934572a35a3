"""reorder_parameters (robosimgs_amd/training.py) on the CPU: pure tensor bookkeeping, no kernel involved."""
import pytest
import torch


def _params(n=500, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"means": torch.randn(n, 3, generator=g).requires_grad_(True), "scales": torch.rand(n, 3, generator=g).requires_grad_(True),
            "colors": torch.randn(n, 4, 3, generator=g).requires_grad_(True)}


def test_reorder_permutes_parameters_optimizer_state_and_extra_state_once():
    from robosimgs_amd.training import reorder_parameters
    p = _params()
    n = p["means"].shape[0]
    p["alias"] = p["scales"]                                  # the same tensor under a second key
    p["lr_table"] = torch.arange(n, dtype=torch.float32)      # N rows by accident: NOT per-Gaussian
    opt = torch.optim.Adam([p["means"], p["scales"], p["colors"]], lr=1e-2)
    (p["means"].sum() + (p["scales"] ** 2).sum() + p["colors"].sum()).backward()
    opt.step()
    before = {k: v.detach().clone() for k, v in p.items()}
    m_before = opt.state[p["scales"]]["exp_avg"].clone()
    grad2d = torch.arange(n, dtype=torch.float32) * 2.0       # a densification strategy's accumulator
    order = reorder_parameters(p, opt, per_gaussian=("means", "scales", "alias", "colors"), extra_state=[grad2d])
    assert sorted(order.tolist()) == list(range(n))
    for k in ("means", "scales", "colors"):
        assert torch.equal(p[k].detach(), before[k][order]), k
    assert p["alias"] is p["scales"]                          # permuted ONCE, not twice
    assert torch.equal(p["lr_table"], before["lr_table"])     # not named: untouched
    assert torch.equal(opt.state[p["scales"]]["exp_avg"], m_before[order])
    assert torch.equal(grad2d, (torch.arange(n, dtype=torch.float32) * 2.0)[order])
    # the default (no key list) still takes everything with N rows, each storage once
    q = _params(seed=1)
    q["alias"] = q["means"]
    b = q["means"].detach().clone()
    o2 = reorder_parameters(q)
    assert torch.equal(q["means"].detach(), b[o2])
    with pytest.raises(ValueError):
        reorder_parameters(_params(), extra_state=[torch.zeros(7)])
