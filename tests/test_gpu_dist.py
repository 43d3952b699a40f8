"""Data-parallel semantics on the GPU with two processes sharing the one device (gloo moves the CUDA tensors
through the host): SyncBatchNorm over two half-batches must equal BatchNorm over the whole batch, and the
bucketed gradient all-reduce + fused Adam must equal a single-process step on the mean gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    torch.cuda.set_device(0)
    import lidog_amd.me as ME
    from lidog_amd.trainer import FlatAdam
    g = torch.Generator().manual_seed(5)
    n, C = 4001, 64
    x = torch.randn(n, C, generator=g) * 1.5 + 0.3
    r = torch.randn(n, C, generator=g)
    gy = torch.randn(n, C, generator=g)
    half = slice(0, 1777) if rank == 0 else slice(1777, n)  # uneven shards: the row count must be all-reduced too
    # reference: plain BN over the whole batch in this process
    ref = ME.MinkowskiBatchNorm(C).cuda()
    xr = x.cuda().requires_grad_(True)
    yr = ME.batch_norm(xr, ref.bn, 1, True, r.cuda(), None)
    yr.backward(gy.cuda())
    # SyncBN over the two shards
    mod = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(ME.MinkowskiBatchNorm(C)).cuda()
    assert isinstance(mod, ME.MinkowskiSyncBatchNorm) and list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    xs = x[half].cuda().requires_grad_(True)
    ys = ME.batch_norm(xs, mod.bn, 1, True, r[half].cuda(), mod._sync_group())
    ys.backward(gy[half].cuda())
    ok = torch.allclose(ys, yr[half], rtol=1e-5, atol=1e-6) and torch.allclose(xs.grad, xr.grad[half], rtol=1e-4, atol=1e-6)
    ok = ok and torch.allclose(mod.bn.running_var, ref.bn.running_var, rtol=1e-5, atol=1e-7)
    # parameter gradients are LOCAL sums; summed over ranks they equal the full-batch gradient
    gw = mod.bn.weight.grad.clone()
    dist.all_reduce(gw)
    ok = ok and torch.allclose(gw, ref.bn.weight.grad, rtol=1e-4, atol=1e-5)

    # gradient buckets + Adam: mean of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Linear(64, 8)).cuda()
    twin = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Linear(64, 8)).cuda()
    twin.load_state_dict(net.state_dict())
    opt = FlatAdam(net, lr=1e-2, weight_decay=1e-4, bucket_bytes=4096)
    ref_opt = torch.optim.Adam(twin.parameters(), lr=1e-2, weight_decay=1e-4)
    data = [torch.randn(16, 32, generator=torch.Generator().manual_seed(10 + k)).cuda() for k in range(world)]
    for _ in range(2):
        opt.zero_grad()
        net(data[rank]).square().mean().backward()
        opt.step()
        ref_opt.zero_grad()
        sum(twin(d).square().mean() for d in data).div(world).backward()
        ref_opt.step()
    for a, b in zip(net.parameters(), twin.parameters()):
        ok = ok and torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_syncbn_and_ddp_two_processes_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}
