"""Failure paths of the data-parallel transports, with two processes sharing the one GPU (gloo for the process group,
hipIpc mailboxes for the peer all-reduce), driven by the fault switches only lidog_amd.comm reads (LIDOG_PEER_FAULT):

* a rank that cannot open its peers' mailboxes -> EVERY rank ends without the peer path, the note names the reason, the
  SyncBatchNorm statistics still come out right through the fallback;
* a rank that never raises one flag -> the error word is set on every rank within the wait limit, no later call waits,
  Transport.check() raises on EVERY rank and the processes leave non-zero;
* ranks with unequal work (scans of +-20 % voxels) through 50 optimiser steps with the peer path on -> no time-out,
  parameters and SyncBatchNorm running statistics bit-identical on both ranks.
Reference: train_lidog.py:227-231,286-301 (DDP + MinkowskiSyncBatchNorm over the GPUs of one node)."""
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO, seeded_state_dict, small_batch

pytestmark = pytest.mark.gpu


def _init(rank, world, port, env):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.cuda.set_device(0)


def _run(target, env, world=2, timeout=600, port_base=30100, expect_exit=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + os.getpid() % 1500
    procs = [ctx.Process(target=target, args=(r, world, port, q, env)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.terminate()
                p.join(30)
    assert len(got) == world, got
    assert all(g[1] for g in got), got
    if expect_exit is not None:
        assert all((p.exitcode == 0) == (expect_exit == 0) for p in procs), [p.exitcode for p in procs]
    return sorted(got)


# ------------------------------------------------------------------ a rank cannot open a mailbox
def _open_fault_worker(rank, world, port, q, env):
    try:
        _init(rank, world, port, env)
        import lidog_amd.me as ME
        from lidog_amd import comm
        tr = comm.transport()
        ok = tr.peer is None and tr.peer_probe is None
        note = tr.peer_note
        ok = ok and (("injected fault" in note) if rank == 1 else ("another rank" in note))
        # the statistics still travel (torch.distributed here) and are right: SyncBN over two uneven shards == BN over all
        g = torch.Generator().manual_seed(5)
        n, C = 3001, 32
        x = torch.randn(n, C, generator=g) * 1.5 + 0.3
        gy = torch.randn(n, C, generator=g)
        half = slice(0, 1301) if rank == 0 else slice(1301, n)
        ref = ME.MinkowskiBatchNorm(C).cuda()
        xr = x.cuda().requires_grad_(True)
        yr = ME.batch_norm(xr, ref.bn, 1, True, None, None)
        yr.backward(gy.cuda())
        mod = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(ME.MinkowskiBatchNorm(C)).cuda()
        xs = x[half].cuda().requires_grad_(True)
        ys = ME.batch_norm(xs, mod.bn, 1, True, None, mod._sync_group())
        ys.backward(gy[half].cuda())
        ok = ok and torch.allclose(ys, yr[half], rtol=1e-5, atol=1e-6) and \
            torch.allclose(xs.grad, xr.grad[half], rtol=1e-4, atol=1e-6)
        tr.check()      # nothing to complain about: the peer path does not exist
        q.put((rank, bool(ok), note))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_a_rank_that_cannot_open_a_mailbox_sends_every_rank_to_the_fallback():
    _run(_open_fault_worker, {"LIDOG_PEER_ALLREDUCE": "1", "LIDOG_PEER_FAULT": "open:1"})


# ------------------------------------------------------------------ a flag that never arrives
def _skip_flag_worker(rank, world, port, q, env):
    try:
        _init(rank, world, port, env)
        from lidog_amd import _lib, comm
        tr = comm.transport()
        if tr.peer is None:
            q.put((rank, False, "peer all-reduce not set up: " + tr.peer_note))
            return
        t0 = time.time()
        for it in range(12):        # call 3 of rank 1 raises no flags
            t = torch.full((193,), float(rank + 1), dtype=torch.float64, device="cuda")
            tr.allreduce_f64(t)
        torch.cuda.synchronize()
        waited = time.time() - t0
        status = _lib.load().lidog_peer_status(tr.peer)
        raised = False
        try:
            tr.check()
        except RuntimeError as e:
            raised = "peer all-reduce failed" in str(e)
        # ONE wait limit (2^19 polls: a second or two), not one per call after the failure
        q.put((rank, bool(raised and status != 0 and waited < 60), f"status {status} waited {waited:.1f}s raised {raised}"))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(1 if raised else 0)      # what a driver does with the exception: leave non-zero
    except SystemExit:
        raise
    except Exception as e:
        import traceback
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_a_flag_that_never_arrives_raises_on_every_rank_within_the_wait_limit():
    got = _run(_skip_flag_worker, {"LIDOG_PEER_ALLREDUCE": "1", "LIDOG_PEER_FAULT": "skipflag:1:3",
                                   "LIDOG_PEER_SPIN_LIMIT": str(1 << 19)}, expect_exit=1)
    assert len(got) == 2


# ------------------------------------------------------------------ unequal work per rank
def _imbalance_worker(rank, world, port, q, env):
    try:
        _init(rank, world, port, env)
        import lidog_amd
        from lidog_amd import comm
        from lidog_amd.optim import FlatAdam
        from lidog_amd.trainer import LiDOGStep, setup_data_parallel
        kw = dict(in_channels=1, out_channels=7, D=3, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
        torch.manual_seed(7 + rank)
        model = lidog_amd.MinkUNet34BEV(**kw)
        model = setup_data_parallel(model.cuda())
        model.train()
        step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4, bucket_bytes=8 << 20))
        # rank 0 sees scans of 1 800 points, rank 1 of 1 200 (+-20 % around 1 500): the ranks finish their kernels at
        # different times all the way through the step, the statistics messages lock them together again 241 times
        n_points = 1800 if rank == 0 else 1200
        batches = []
        for i in range(2):
            coords = small_batch((50 + 2 * i + rank,), n_points=n_points)
            g = torch.Generator().manual_seed(900 + 2 * i + rank)
            batches.append({"coords_int": coords.cuda(), "source_features0": torch.ones((coords.shape[0], 1), device="cuda"),
                            "source_sem_labels0": torch.randint(-1, 7, (coords.shape[0],), generator=g).cuda(),
                            "source_bev_labels0": {"block8": torch.randint(-1, 7, (1, 17, 17), generator=g).cuda()}})
        rows = [b["coords_int"].shape[0] for b in batches]
        t0 = time.time()
        for i in range(6):
            out = step.training_step(batches[i % 2])
        torch.cuda.synchronize()
        dt = time.time() - t0
        tr = comm.transport()
        tr.check()                                  # collective; raises on a time-out anywhere
        flat = step.opt.flat.flat
        stats = torch.cat([b.detach().float().flatten() for n, b in model.named_buffers()
                           if "running" in n and not n.startswith("encoders2d")])
        same = True
        for t in (flat, stats):
            ref = t.detach().clone()
            dist.broadcast(ref, src=0)
            same = same and bool(torch.equal(ref.view(torch.int32), t.detach().view(torch.int32)))
        finite = bool(torch.isfinite(out["loss"]).item())
        q.put((rank, bool(same and finite and tr.peer is not None),
               f"rows {rows} peer {tr.peer_note} identical {same} loss {float(out['loss']):.4f} {dt:.1f}s"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_ranks_with_unequal_work_stay_bit_identical_over_six_steps():
    """(50 steps until round 5: 158 s of the suite; a rank that falls out of step shows in the first steps -- the ticket bug of
    round 5 showed in step 1 -- and the 80-step soak is scripts/soak_side_streams.py, profiles/r05_soak_side_streams.txt)"""
    got = _run(_imbalance_worker, {"LIDOG_PEER_ALLREDUCE": "1"}, timeout=900)
    rows0, rows1 = (eval(g[2].split("rows ")[1].split(" peer")[0]) for g in got)
    assert min(rows0) > 1.15 * max(rows1), (rows0, rows1)
