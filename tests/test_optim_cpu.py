"""Host logic of lidog_amd.optim on the CPU: learning-rate schedules against torch's own schedulers, the
DistributedSampler-style sharding, once-per-backward handout of flat gradient slices, skip rule of the optimisers."""
import math

import pytest
import torch
import torch.nn as nn

import helpers  # noqa: F401  (sys.path)


class _FakeOpt:
    def __init__(self, lr):
        self.lr = self.base_lr = lr


@pytest.mark.parametrize("name", ["CosineAnnealingLR", "ExponentialLR", "CyclicLR"])
@pytest.mark.parametrize("lr", [1e-3, 1e-2])
def test_schedules_equal_torch(name, lr):
    """the lr sequence over 45 epochs is EXACTLY torch's (hyper-parameters of trainer_lighting_2d.py:379-389)"""
    from lidog_amd.optim import make_scheduler
    p = nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=lr)
    if name == "CosineAnnealingLR":
        ts = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=10)
    elif name == "ExponentialLR":
        ts = torch.optim.lr_scheduler.ExponentialLR(topt, gamma=0.99)
    else:
        ts = torch.optim.lr_scheduler.CyclicLR(topt, base_lr=lr / 10000, max_lr=lr, step_size_up=5, mode="triangular2",
                                               cycle_momentum=False)
    opt = _FakeOpt(lr)
    s = make_scheduler(name, opt)
    for epoch in range(45):
        assert opt.lr == topt.param_groups[0]["lr"], (name, epoch, opt.lr, topt.param_groups[0]["lr"])
        topt.step()
        ts.step()
        s.step()
    sd = s.state_dict()
    opt2 = _FakeOpt(lr)
    s2 = make_scheduler(name, opt2)
    s2.load_state_dict(sd)
    opt2.lr = opt.lr
    for _ in range(5):
        s.step(), s2.step(), ts.step()
        assert opt2.lr == opt.lr == topt.param_groups[0]["lr"]


def test_unknown_names_raise_like_the_reference():
    from lidog_amd.optim import make_scheduler, make_optimizer
    with pytest.raises(NotImplementedError):
        make_scheduler("StepLR", _FakeOpt(1e-3))
    with pytest.raises(NotImplementedError):
        make_optimizer("RMSprop", nn.Linear(2, 2), 1e-3)
    assert make_scheduler(None, _FakeOpt(1e-3)) is None


@pytest.mark.parametrize("n,world", [(10, 4), (16, 8), (3, 8), (128, 4), (7, 2)])
@pytest.mark.parametrize("shuffle", [False, True])
def test_shard_indices_equal_distributed_sampler(n, world, shuffle):
    from torch.utils.data import DistributedSampler
    from lidog_amd.optim import shard_indices
    data = list(range(n))
    for epoch in (0, 3):
        seen = []
        for rank in range(world):
            ds = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=shuffle, seed=1234)
            ds.set_epoch(epoch)
            want = list(iter(ds))
            got = shard_indices(n, rank, world, shuffle=shuffle, seed=1234, epoch=epoch)
            assert got == want, (n, world, rank, epoch)
            seen += got
        assert len(seen) == world * math.ceil(n / world) and set(seen) == set(data)   # nobody runs dry


def _tiny():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(4, 3), nn.Linear(3, 2))


def test_flat_slice_is_handed_out_once_per_backward():
    """two autograd nodes using one parameter must not both write the same flat slice"""
    from lidog_amd.optim import FlatParams
    from lidog_amd.me import _grad_out
    m = _tiny()
    flat = FlatParams(m)
    p = m[0].weight
    flat.zero_grad()
    a = _grad_out(p, p.shape)
    assert a is not None and a.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[0]
    assert _grad_out(p, p.shape) is None            # second node of the same pass: fresh tensor, autograd accumulates
    flat.zero_grad()
    assert _grad_out(p, p.shape) is not None        # next pass: handed out again
    p.grad = torch.zeros_like(p)
    flat.generation += 1
    assert _grad_out(p, p.shape) is None            # a .grad already present: autograd must accumulate


def test_two_uses_of_one_parameter_accumulate_correctly():
    """CPU analogue of calling the model twice before one backward: the node that gets the flat view writes it, the
    other returns a fresh tensor; the sum must equal plain autograd's"""
    from lidog_amd.optim import FlatParams
    from lidog_amd.me import _grad_out

    class Mul(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.w = w
            return x * w

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            gw = _grad_out(ctx.w, ctx.w.shape)
            val = (g * x).sum(0)
            if gw is None:
                gw = val
            else:
                gw.copy_(val)          # overwrites, as the HIP kernels do
            return g * ctx.w, gw

    lin = nn.Linear(3, 3)
    w = nn.Parameter(torch.randn(3))
    holder = nn.ParameterList([w])
    flat = FlatParams(holder)
    x1, x2 = torch.randn(5, 3), torch.randn(5, 3)
    flat.zero_grad()
    (Mul.apply(x1, w).sum() + Mul.apply(x2, w).pow(2).sum()).backward()
    flat.gather_strays()
    got = flat.grad.clone()
    w2 = w.detach().clone().requires_grad_(True)
    ((x1 * w2).sum() + (x2 * w2).pow(2).sum()).backward()
    torch.testing.assert_close(got, w2.grad)
    del lin


def test_optimizer_runs_skip_parameters_without_gradient():
    from lidog_amd.optim import _FlatOptimizer
    m = _tiny()
    opt = _FlatOptimizer(m, 1e-3)
    opt.zero_grad()
    params = opt.flat.params
    for p in params[:2]:
        p.grad = torch.zeros_like(p)
    runs = opt._runs()
    assert runs == [[0, params[0].numel() + params[1].numel(), 1]]
    assert opt.param_steps == [1, 1, 0, 0]
    for p in params:
        p.grad = torch.zeros_like(p)
    runs = opt._runs()     # first two are at step 2, the others at step 1: two kernel launches
    n01 = params[0].numel() + params[1].numel()
    assert runs == [[0, n01, 2], [n01, opt.flat.total, 1]]


@pytest.mark.parametrize("kind", ["Adam", "SGD"])
def test_optimizer_state_interchanges_with_torch_optim(kind):
    """Lightning stores torch.optim's own state layout in `optimizer_states` (trainer.fit(ckpt_path=...),
    train_lidog.py:298-301).  The flat optimisers write that layout (checkpoint.save_lightning_checkpoint) and read it:
    torch -> flat -> torch is the identity, a parameter that never got a gradient has no entry on either side."""
    import copy
    from lidog_amd.optim import FlatAdam, FlatSGD
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    twin = copy.deepcopy(net)
    if kind == "Adam":
        topt = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=1e-4)
    else:
        topt = torch.optim.SGD(twin.parameters(), lr=1e-2, momentum=0.98, weight_decay=1e-4, nesterov=True)
    for it in range(3):     # the last layer never takes part: no gradient, no state
        topt.zero_grad()
        twin[1](twin[0](torch.randn(7, 6))).square().mean().backward()
        topt.step()
    sd = copy.deepcopy(topt.state_dict())
    opt = (FlatAdam(net, lr=0.5, weight_decay=1e-4) if kind == "Adam" else FlatSGD(net, lr=0.5, weight_decay=1e-4))
    opt.load_state_dict(sd)
    # torch's SGD keeps no step count: a parameter with a momentum buffer counts as stepped once
    assert opt.lr == 1e-2 and opt.param_steps == ([3, 3, 3, 3, 0, 0] if kind == "Adam" else [1, 1, 1, 1, 0, 0])
    names = ("exp_avg", "exp_avg_sq") if kind == "Adam" else ("momentum_buffer",)
    for i, (p, off) in enumerate(zip(opt.flat.params, opt.flat.offsets)):
        for k in names:
            got = getattr(opt, k)[off:off + p.numel()].view(p.shape)
            want = sd["state"][i][k] if i in sd["state"] else torch.zeros_like(p)
            assert torch.equal(got, want), (i, k)
    back = opt.torch_state_dict()
    assert sorted(back["state"]) == sorted(sd["state"]) == [0, 1, 2, 3]
    fresh = (torch.optim.Adam(twin.parameters(), lr=1.0) if kind == "Adam" else
             torch.optim.SGD(twin.parameters(), lr=1.0, momentum=0.98, nesterov=True))
    fresh.load_state_dict(back)            # torch accepts what we write
    for i in back["state"]:
        for k in names:
            assert torch.equal(back["state"][i][k], sd["state"][i][k])
        if kind == "Adam":
            assert float(back["state"][i]["step"]) == float(sd["state"][i]["step"]) == 3.0
    g = fresh.param_groups[0]
    assert g["lr"] == 1e-2 and g["weight_decay"] == 1e-4 and back["param_groups"][0]["params"] == list(range(6))
