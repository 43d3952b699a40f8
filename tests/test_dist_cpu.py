"""world_size-2 gloo tests of the data-parallel host logic (gradient buckets, sampler sharding)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    from lidog_amd.trainer import FlatParams, GradientBuckets, shard_indices
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                                torch.nn.Linear(256, 7))
    flat = FlatParams(model)
    buckets = GradientBuckets(flat, bucket_bytes=8 * 1024)  # several buckets
    assert len(buckets.slices) >= 3
    covered = sorted(buckets.slices)
    assert covered[0][0] == 0 and covered[-1][1] == flat.total
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    x = torch.randn(32, 64, generator=torch.Generator().manual_seed(100 + rank))
    for it in range(2):  # twice: bucket counters must re-arm
        flat.zero_grad()
        model(x).square().mean().backward()
        buckets.finish()
        # reference: explicit all-reduce of independently computed local grads
        ref_model = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256),
                                        torch.nn.ReLU(), torch.nn.Linear(256, 7))
        ref_model.load_state_dict(model.state_dict())
        ref_model(x).square().mean().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in ref_model.parameters()])
        dist.all_reduce(ref)
        assert torch.allclose(flat.grad, ref, rtol=1e-6, atol=1e-7), (flat.grad - ref).abs().max()
    idx = shard_indices(11, rank, world)
    # DDP's start-up broadcast: ranks that seeded differently end up with rank 0's parameters
    from lidog_amd.optim import _FlatOptimizer
    torch.manual_seed(100 + rank)
    net = torch.nn.Linear(8, 8)
    opt = _FlatOptimizer(net, 1e-3)
    mine = opt.flat.flat.clone()
    dist.broadcast(mine, src=0)
    assert torch.equal(mine, opt.flat.flat) and net.weight.data_ptr() == opt.flat.flat.data_ptr()
    q.put((rank, idx))
    dist.destroy_process_group()


def test_gradient_buckets_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # DistributedSampler semantics: strided, padded by wrap-around so that both ranks get the same count
    assert got[0] == [0, 2, 4, 6, 8, 10] and got[1] == [1, 3, 5, 7, 9, 0]
