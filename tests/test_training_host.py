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


def test_trainer_reorders_by_itself_and_follows_the_unreordered_trajectory():
    """Trainer (robosimgs_amd/training.py) with a toy, permutation-equivariant render function on the CPU: a loop that lets
    the trainer reorder every 3 steps ends where the plain loop ends (fp64: the sums over the Gaussians are the only thing
    whose order changes), original_index maps back, extra state follows, rebind() resets."""
    from robosimgs_amd.training import Trainer
    torch.manual_seed(0)
    n = 300

    def make():
        g = torch.Generator().manual_seed(3)
        P = {"means": torch.randn(n, 3, generator=g, dtype=torch.float64), "quats": torch.randn(n, 4, generator=g, dtype=torch.float64),
             "scales": torch.rand(n, 3, generator=g, dtype=torch.float64), "opacities": torch.rand(n, generator=g, dtype=torch.float64),
             "colors": torch.randn(n, 4, 3, generator=g, dtype=torch.float64)}
        return {k: v.requires_grad_(True) for k, v in P.items()}

    def toy_render(means, quats, scales, opacities, colors, viewmats, Ks, width, height, gain=1.0):
        w = torch.sigmoid(means @ viewmats[0, :3, :3].T).sum(-1) * opacities                      # per Gaussian
        img = (w[:, None] * colors[:, 0] * scales).sum(0) * gain + (quats ** 2).sum()
        return img[None], w.sum()[None], {"n": means.shape[0]}

    vm = torch.eye(4, dtype=torch.float64)[None]
    target = torch.tensor([0.3, -0.2, 0.9], dtype=torch.float64)
    # plain loop
    A = make()
    optA = torch.optim.Adam(A.values(), lr=1e-2)
    for _ in range(8):
        c, a, _m = toy_render(*[A[k] for k in Trainer.KEYS], vm, None, 4, 4, gain=2.0)
        ((c[0] - target).abs().sum() + 0.1 * a.sum()).backward()
        optA.step(); optA.zero_grad(set_to_none=True)
    # the trainer's loop
    B = make()
    tag = torch.arange(n, dtype=torch.float64)
    tr = Trainer(B, torch.optim.Adam(B.values(), lr=1e-2), 4, 4, auto_reorder_every=3, extra_state=[tag], render_fn=toy_render, gain=2.0)
    for _ in range(8):
        c, a, m = tr.render(vm, None)
        tr.step((c[0] - target).abs().sum() + 0.1 * a.sum())
    assert tr.reorders == 3 and tr.it == 8                       # steps 0, 3, 6
    assert sorted(tr.original_index.tolist()) == list(range(n)) and not torch.equal(tr.original_index, torch.arange(n))
    assert torch.equal(tag, tr.original_index.to(tag.dtype))     # extra state followed
    for k in Trainer.KEYS:
        torch.testing.assert_close(tr.in_original_order(B[k].detach()), A[k].detach(), rtol=1e-9, atol=1e-11)
    # rebind: new tensors, fresh identity, reorder due again
    C = make()
    tr.rebind(C, None)
    assert torch.equal(tr.original_index, torch.arange(n))
    tr.render(vm, None)
    assert tr.reorders == 4
    with pytest.raises(KeyError):
        Trainer({"means": C["means"]}, None, 4, 4)
