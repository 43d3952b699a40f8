"""Step driver and owning training loop on the GPU: fused optimisers against
torch.optim, the reference's recipe (Adam 0.01 + ExponentialLR, configs/source/single/semantickitti.yaml:38-46) through
lidog_amd.train.Fit against the CPU oracle driven by torch.optim + torch's scheduler, checkpoint resume, a model called
twice before one backward, and the full LiDOG step on two data-parallel ranks against one rank on the joint batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO, seeded_state_dict, small_batch

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ fused optimisers == torch.optim
def _two_nets():
    torch.manual_seed(3)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32),
                                     torch.nn.Tanh(), torch.nn.Linear(32, 4)).cuda()
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize("kind", ["Adam", "SGD"])
def test_fused_optimizer_equals_torch_with_an_unused_parameter(kind):
    """6 steps; the middle layer gets NO gradient in steps 0-1 (as `final.*` during warm-up epochs): torch skips such
    parameters entirely (no weight decay, no moment update, own step count) and so must the fused kernels"""
    from lidog_amd.optim import make_optimizer
    net, twin = _two_nets()
    opt = make_optimizer(kind, net, 1e-2, weight_decay=1e-4, momentum=0.98)
    if kind == "Adam":
        ref = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=1e-4)
    else:
        ref = torch.optim.SGD(twin.parameters(), lr=1e-2, momentum=0.98, weight_decay=1e-4, nesterov=True)
    g = torch.Generator().manual_seed(0)
    for it in range(6):
        x = torch.randn(8, 16, generator=g).cuda()
        for m, o, zero in ((net, opt, opt.zero_grad), (twin, ref, lambda: ref.zero_grad(set_to_none=True))):
            zero()
            if it < 2:   # skip the middle layer: first -> (tanh) -> last through a fixed random projection
                h = torch.tanh(m[0](x))
                y = m[4](torch.tanh(h))
            else:
                y = m(x)
            y.square().mean().backward()
            o.step()
        for (n, p), q in zip(net.named_parameters(), twin.parameters()):
            torch.testing.assert_close(p, q, rtol=2e-6, atol=2e-7, msg=lambda s: f"{kind} step {it} {n}: {s}")
    assert opt.param_steps == [6, 6, 4, 4, 6, 6]


# ------------------------------------------------------------------ the reference's recipe through Fit
class _Scenes:
    """small synthetic scenes as a dataset: scan i = helpers.small_scene(seed0 + i)"""

    def __init__(self, n, seed0=40, n_points=1200, bev=17):
        self.n, self.seed0, self.n_points, self.bev = n, seed0, n_points, bev

    def __len__(self):
        return self.n

    def batch(self, indices, device):
        coords = small_batch(tuple(self.seed0 + i for i in indices), n_points=self.n_points)
        labels, bev = [], []
        for b, i in enumerate(indices):     # labels belong to the SCAN, whatever batch it is collated into
            g = torch.Generator().manual_seed(1000 + self.seed0 + i)
            labels.append(torch.randint(-1, 7, (int((coords[:, 0] == b).sum()),), generator=g))
            bev.append(torch.randint(-1, 7, (1, self.bev, self.bev), generator=g))
        labels, bev = torch.cat(labels), torch.cat(bev)
        return {"coords_int": coords.to(device), "source_coordinates0": coords.float().to(device),
                "source_features0": torch.ones((coords.shape[0], 1), device=device),
                "source_sem_labels0": labels.to(device), "source_bev_labels0": {"block8": bev.to(device)}}


def _oracle_run(sd, data, epochs, bs, lr, max_steps=None):
    """the same loop on the CPU oracle with torch.optim.Adam + torch's ExponentialLR stepped per epoch; `max_steps`: stop
    computing losses after that many steps (the lr sequence still covers every epoch)"""
    import oracle.me_cpu as OME
    from oracle.ref_torch import soft_dice_loss_ref
    from lidog_amd.minkunet import make_models
    from lidog_amd.optim import shard_indices
    OME.set_mode("blas")
    model = make_models(OME).MinkUNet34(in_channels=1, out_channels=7, D=3)
    model.load_state_dict(sd)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.99)
    losses, lrs = [], []
    for epoch in range(epochs):
        idx = shard_indices(len(data), 0, 1, shuffle=True, seed=1234, epoch=epoch)
        for i in range(0, len(idx), bs):
            if max_steps is not None and len(losses) >= max_steps:
                break
            b = data.batch(idx[i:i + bs], "cpu")
            out = model(OME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]), is_seg=True)
            loss = soft_dice_loss_ref(out.F, b["source_sem_labels0"])
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    OME.set_mode("exact")
    return losses, lrs


def test_fit_source_recipe_matches_oracle_and_resumes(tmp_path):
    """2 epochs x 3 steps of config 1's recipe (MinkUNet34, SoftDICE, Adam lr 0.01 + ExponentialLR 0.99, reshuffled
    every epoch): loss trajectory against the oracle, lr sequence EXACT, per-epoch checkpoints, validation cadence,
    and a run resumed from the epoch-0 checkpoint reproduces epoch 1"""
    import lidog_amd
    from lidog_amd.train import Fit, last_checkpoint
    data, val = _Scenes(6), _Scenes(4, seed0=80)
    proto = lidog_amd.MinkUNet34(in_channels=1, out_channels=7, D=3)
    sd = seeded_state_dict(proto, seed=11)
    kw = dict(model_kind="MinkUNet34", batch_size=2, optimizer="Adam", lr=1e-2, scheduler="ExponentialLR",
              train_data=data, val_data=val, check_val_every_n_epoch=2, state_dict=sd, log=lambda *_: None)
    fit = Fit(epochs=2, save_dir=str(tmp_path / "a"), **kw)
    hist = fit.run()
    assert [h["global_step"] for h in hist] == [3, 6]
    assert "validation" not in hist[0] and hist[1]["validation"]["steps"] == 2     # check_val_every_n_epoch=2
    assert all(os.path.exists(h["checkpoint"]) for h in hist)                      # every_n_epochs=1, keep all
    assert last_checkpoint(str(tmp_path / "a")) == hist[1]["checkpoint"]
    # the oracle runs epoch 0 and the first step of epoch 1 (reshuffled order, decayed lr): 4 of the 6 steps
    ref_losses, ref_lrs = _oracle_run(sd, data, 2, 2, 1e-2, max_steps=4)
    assert [h["lr"] for h in hist] == ref_lrs                                      # 0.01, 0.01 * 0.99 exactly
    got = (hist[0]["losses"] + hist[1]["losses"])[:4]
    err = np.abs(np.array(got) - np.array(ref_losses))
    # exact before the first update, 1e-4 after one (north_star bar); Adam's first updates are +-lr*sign(g) per
    # element (lr 0.01 here), so near-zero gradient entries separate the two trajectories from then on
    assert err[0] <= 1e-5 and err[1] <= 1e-4 and err.max() <= 5e-2, (err, got, ref_losses)
    # Lightning checkpoint layout + resume
    ck = torch.load(hist[0]["checkpoint"], map_location="cpu", weights_only=False)
    assert ck["epoch"] == 0 and ck["global_step"] == 3 and all(k.startswith("model.") for k in ck["state_dict"])
    # scheduler position + the optimiser state in torch.optim's layout (what Lightning stores and restores)
    assert ck["lr_schedulers"][0]["last_epoch"] == 1 and set(ck["optimizer_states"][0]) == {"state", "param_groups"}
    again = Fit(epochs=2, save_dir=str(tmp_path / "b"), resume=hist[0]["checkpoint"], **kw)
    assert again.epoch == 1 and again.global_step == 3 and again.opt.lr == ref_lrs[1]
    h2 = again.run()
    assert len(h2) == 1 and h2[0]["epoch"] == 1 and h2[0]["global_step"] == 6
    np.testing.assert_allclose(h2[0]["losses"], hist[1]["losses"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(h2[0]["validation"]["sem_loss"], hist[1]["validation"]["sem_loss"], rtol=0, atol=1e-6)


def test_fit_lidog_sgd_cosine_runs_with_warmup(tmp_path):
    """MinkUNet34BEV, SGD nesterov 0.98 + CosineAnnealingLR, one warm-up epoch (semantic loss off, `final.*` without a
    gradient: trainer_lighting_2d.py:193-201): final.* must stay untouched during warm-up and move afterwards"""
    import lidog_amd
    from lidog_amd.train import Fit
    proto = lidog_amd.MinkUNet34BEV(in_channels=1, out_channels=7, D=3, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
    sd = seeded_state_dict(proto, seed=5)
    fit = Fit(model_kind="MinkUNet34BEV", bound_2d=5.0, batch_size=2, optimizer="SGD", lr=1e-3,
              scheduler="CosineAnnealingLR", epochs=2, warmup_epochs=1, train_data=_Scenes(4), state_dict=sd,
              log=lambda *_: None)
    w0 = fit.model.final.kernel.detach().clone()
    e0 = fit.model.encoders2d["block8"].out_conv.conv.weight.detach().clone()
    fit.epochs = 1
    fit.run()
    assert torch.equal(fit.model.final.kernel, w0), "a parameter without gradient was updated (weight decay / momentum)"
    assert not torch.equal(fit.model.encoders2d["block8"].out_conv.conv.weight, e0)
    fit.epochs = 2
    h = fit.run()
    assert not torch.equal(fit.model.final.kernel, w0)
    import math
    assert h[-1]["lr"] == pytest.approx(1e-3 * (1 + math.cos(math.pi / 10)) / 2, rel=1e-12) and all(np.isfinite(x) for x in h[-1]["losses"])


# ------------------------------------------------------------------ one parameter, two autograd nodes
def test_model_called_twice_before_one_backward():
    """trainer_lighting_2d_multi.py:166-167 calls the model on two batches and runs ONE backward: every parameter is
    then used by two autograd nodes, which must not both write the same flat gradient slice"""
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd.losses import SoftDICELoss
    from lidog_amd.optim import FlatAdam
    model = lidog_amd.MinkUNet34(in_channels=1, out_channels=7, D=3)
    model.load_state_dict(seeded_state_dict(model, seed=9))
    model.cuda().train()
    opt = FlatAdam(model, lr=1e-3)
    data = _Scenes(2, seed0=60)
    b0, b1 = data.batch([0], "cuda"), data.batch([1], "cuda")
    crit = SoftDICELoss(ignore_label=-1)

    def loss_of(b):
        out = model(ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]), is_seg=True)
        return crit(out.F, b["source_sem_labels0"])

    def snapshot_bn():
        return {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}

    bn0 = snapshot_bn()
    grads = []
    for b in (b0, b1):      # separately: one use per parameter
        model.load_state_dict(bn0, strict=False)
        opt.zero_grad()
        loss_of(b).backward()
        opt.flat.gather_strays()
        torch.cuda.synchronize()
        grads.append(opt.flat.grad.clone())
    model.load_state_dict(bn0, strict=False)
    opt.zero_grad()
    (loss_of(b0) + loss_of(b1)).backward()
    opt.flat.gather_strays()
    torch.cuda.synchronize()
    want = grads[0] + grads[1]
    got = opt.flat.grad
    assert torch.isfinite(got).all()
    for p, off in zip(opt.flat.params, opt.flat.offsets):
        a, b = got[off:off + p.numel()], want[off:off + p.numel()]
        assert (a - b).norm() <= 1e-5 * b.norm() + 1e-9, (off, float((a - b).norm()), float(b.norm()))


# ------------------------------------------------------------------ two data-parallel ranks == one rank, joint batch
def _dp_worker(rank, world, port, q, executor=True):
    try:
        _dp_worker_body(rank, world, port, q, executor)
    except Exception as e:   # report instead of dying silently: the peer would sit in its barrier until the timeout
        import traceback
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def _dp_worker_body(rank, world, port, q, executor):
    # LIDOG_PEER_ALLREDUCE=1: the statistics messages take the one-shot peer all-reduce (opt-in for the step)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LIDOG_PEER_ALLREDUCE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.cuda.set_device(0)
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd import bev as BEV
    from lidog_amd.losses import DICELoss, SoftDICELoss
    from lidog_amd.optim import FlatAdam
    from lidog_amd.trainer import LiDOGStep, setup_data_parallel
    from lidog_amd import trunk
    trunk.set_enabled(executor)
    kw = dict(in_channels=1, out_channels=7, D=3, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
    torch.manual_seed(100 + rank)          # ranks seed differently: the start-up broadcast must align them
    model = lidog_amd.MinkUNet34BEV(**kw)
    sd = seeded_state_dict(model, seed=5)
    if rank == 0:
        model.load_state_dict(sd)
    model = setup_data_parallel(model.cuda())
    model.train()
    assert sum(isinstance(m, ME.MinkowskiSyncBatchNorm) for m in model.modules()) == 62
    opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4, bucket_bytes=8 << 20)
    step = LiDOGStep(model, opt)
    data = _Scenes(2, seed0=70, n_points=1500)
    mine = data.batch([rank], "cuda")
    total, sem_l, bev_l, sem = step.forward_loss(mine)
    # a data-parallel rank (SyncBatchNorm + gradient buckets) runs the trunk executor, its collectives issued from C
    # (here through the host callback into torch.distributed: gloo), unless the executor is switched off
    took = type(sem.F.grad_fn).__name__
    assert (took == "_TrunkFnBackward") == executor, took
    from lidog_amd.comm import transport
    peer_note = transport().peer_note         # "on": the statistics went through the one-shot peer all-reduce
    opt.zero_grad()
    total.backward()
    opt._prepare()                           # joins the lane, waits for the buckets: flat.grad = SUM over ranks
    torch.cuda.synchronize()
    assert opt.strays == 0                   # every gradient was written in place (no copy into the flat buffer)
    base = opt.flat.grad.data_ptr()
    assert all(p.grad is not None and p.grad.data_ptr() == base + 4 * off for p, off in zip(opt.flat.params, opt.flat.offsets))
    g_dp = opt.flat.grad / world
    early = opt.buckets.issued_early
    losses = torch.tensor([float(total.detach())], dtype=torch.float64)
    dist.all_reduce(losses)
    ok, msg = True, ""
    if rank == 0:
        # one process, both scans in one batch, plain BatchNorm over all rows (== SyncBN statistics); the two
        # BatchNorm2d of Encoder2D are per-rank in data-parallel runs (as in the reference), i.e. per scan here
        ref = lidog_amd.MinkUNet34BEV(**kw)
        ref.load_state_dict(sd)
        ref.cuda().train()
        ropt = FlatAdam(ref, lr=1e-3, weight_decay=1e-4, local=True)
        both = data.batch([0, 1], "cuda")
        x = ME.SparseTensor(coordinates=both["coords_int"], features=both["source_features0"])
        out, _, levels, seg = ref._trunk_forward(x)
        logits = (seg if seg is not None else ref.final(out)).F
        lv = levels["block8"]
        sem_c, bev_c = SoftDICELoss(ignore_label=-1), DICELoss(ignore_label=-1)
        tot = 0.0
        for b in range(2):
            rows = (lv.C[:, 0] == b).nonzero().flatten()
            cb = lv.C[rows].clone()
            cb[:, 0] = 0
            img = BEV._Sparse2SuperFn.apply(lv.F[rows], cb.contiguous(), 1, 5.0, 0.05, (5, 3, 1))
            pred = ref.encoders2d["block8"](img)
            l_bev = bev_c(pred.view(-1, 7), both["source_bev_labels0"]["block8"][b].view(-1))
            l_sem = sem_c(logits[rows], both["source_sem_labels0"][rows])
            tot = tot + (0.5 * l_sem + 0.5 * l_bev) / 2
        ropt.zero_grad()
        tot.backward()
        ropt.flat.gather_strays()
        torch.cuda.synchronize()
        g_ref = ropt.flat.grad
        if abs(float(losses) / world - float(tot)) > 1e-5:
            ok, msg = False, f"loss {float(losses) / world} vs {float(tot)}"
        d = (sem.F.detach() - logits[(x.C[:, 0] == 0)].detach()).abs().max().item()
        if d > 1e-5:
            ok, msg = False, msg + f" logits differ by {d}"
        worst = 1.0
        for (n, p), off in zip(ref.named_parameters(), ropt.flat.offsets):
            a, b = g_dp[off:off + p.numel()].double(), g_ref[off:off + p.numel()].double()
            c = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
            if c < worst:
                worst, wn = c, n
        if worst < 1 - 1e-5:
            ok, msg = False, msg + f" gradient cosine {worst} at {wn}"
        if early < 1:
            ok, msg = False, msg + " no gradient bucket was reduced during backward"
        if peer_note != "on":
            ok, msg = False, msg + f" peer all-reduce: {peer_note}"
    q.put((rank, ok, msg))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("executor", [True, False], ids=["trunk_executor", "operator_path"])
def test_lidog_step_two_ranks_equal_one_rank_on_the_joint_batch(executor):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 2000 + (0 if executor else 7)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, executor)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    failed = not all(ok for _, ok, _ in got)
    for p in procs:
        p.join(10 if failed else 120)
        if p.is_alive():     # a rank whose peer failed waits in the barrier: end exactly that process
            p.terminate()
            p.join(30)
    assert not failed, got
    assert all(p.exitcode == 0 for p in procs)


def test_train_cli_runs_checkpoints_and_resumes(tmp_path):
    """`python -m lidog_amd.train` (the owning driver's main(), train_source.py's recipe on configs[0] scans): two
    epochs, a checkpoint per epoch with Lightning's keys, --auto-resume picks the last one up; the optimiser state in
    the checkpoint is torch.optim's own layout and loads into torch.optim.Adam over the model's parameters"""
    import subprocess
    import lidog_amd
    cmd = [sys.executable, "-m", "lidog_amd.train", "--model", "MinkUNet34", "--config", "source8k", "--scans", "4",
           "--batch", "2", "--lr", "0.01", "--scheduler", "ExponentialLR", "--save-dir", str(tmp_path)]
    env = dict(os.environ, PYTHONPATH=REPO)
    out = subprocess.run(cmd + ["--epochs", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    ckpts = sorted(os.listdir(os.path.join(str(tmp_path), "checkpoints")))
    assert ckpts == ["epoch=0-step=2.ckpt", "epoch=1-step=4.ckpt"], ckpts
    ck = torch.load(os.path.join(str(tmp_path), "checkpoints", ckpts[-1]), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and ck["global_step"] == 4 and all(k.startswith("model.") for k in ck["state_dict"])
    osd = ck["optimizer_states"][0]
    assert set(osd) == {"state", "param_groups"} and ck["lr_schedulers"][0]["last_epoch"] == 2
    model = lidog_amd.MinkUNet34(1, 7, 3)
    topt = torch.optim.Adam(model.parameters(), lr=1.0)
    topt.load_state_dict(osd)                         # what Lightning's fit(ckpt_path=...) does on the reference side
    assert abs(topt.param_groups[0]["lr"] - 0.01 * 0.99 ** 2) < 1e-12
    assert len(topt.state) == len(list(model.parameters())) and float(next(iter(topt.state.values()))["step"]) == 4.0
    out = subprocess.run(cmd + ["--epochs", "3", "--auto-resume"], env=env, capture_output=True, text=True, timeout=600,
                         cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    assert sorted(os.listdir(os.path.join(str(tmp_path), "checkpoints")))[-1] == "epoch=2-step=6.ckpt"
