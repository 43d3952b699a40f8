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


def _bench_line(extra_env):
    import json
    import subprocess
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_single_rank_rccl_runs_every_data_parallel_path():
    """The collectives of the N > 1 runs against the real RCCL library (backend "nccl") in a ONE-rank process
    group: fp64 SyncBN statistics all-reduces (forward and backward of the 62 sparse BNs), bucketed asynchronous
    gradient all-reduces issued from the second backward stream, barrier, destroy.  With one rank every
    all-reduce is the identity, so the step must reproduce the plain single-GPU step."""
    port = 29700 + os.getpid() % 2000
    # (LIDOG_WGRAD_FIT=0 on both sides: by default the in-line weight gradients of the plain run are cut into different
    # work items than the second stream's, me._wgrad_chunk -- other partial sums, and Adam amplifies the last bits)
    plain = _bench_line({"LIDOG_BACKWARD_OVERLAP": "0", "LIDOG_WGRAD_FIT": "0"})
    dp = _bench_line({"LIDOG_BENCH_SINGLE_RANK_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                      "LIDOG_WGRAD_FIT": "0"})
    assert dp["config"]["parallelism"] == "dp1+syncbn" and plain["config"]["parallelism"] == "dp1"
    assert abs(dp["loss"] - plain["loss"]) <= 2e-5 * abs(plain["loss"]), (dp["loss"], plain["loss"])
    assert dp["config"]["collectives"] == "native" and dp["mode"] == "normal"
    # the safe mode bench.py falls back to after a hung N > 1 run (LIDOG_DP_SAFE=1: torch.distributed's communicator for
    # statistics and buckets, buckets after backward, no side streams) computes the same step
    safe = _bench_line({"LIDOG_BENCH_SINGLE_RANK_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port + 1),
                        "LIDOG_WGRAD_FIT": "0", "LIDOG_DP_SAFE": "1"})
    assert safe["config"]["collectives"] == "torch" and "dp_safe" in safe["config"]
    assert safe["config"]["trunk_path"] == "executor" and safe["replicas_identical"]["parameters"] is True
    assert abs(safe["loss"] - plain["loss"]) <= 2e-5 * abs(plain["loss"]), (safe["loss"], plain["loss"])


def _rccl_c_abi_worker(q):
    """own process: RCCL initialised through the C ABI only (no torch.distributed)"""
    import ctypes
    sys.path.insert(0, REPO)
    torch.cuda.set_device(0)
    from lidog_amd import _lib
    L = _lib.load()
    nbytes = L.lidog_comm_unique_id_bytes()
    uid = ctypes.create_string_buffer(nbytes)
    assert L.lidog_comm_unique_id(uid) == 0, L.lidog_last_error()
    comm = ctypes.c_void_p()
    assert L.lidog_comm_init_rank(uid, 1, 0, ctypes.byref(comm)) == 0, L.lidog_last_error()
    x = torch.randn(1 << 20, device="cuda")
    s = torch.randn(193, device="cuda", dtype=torch.float64)
    x0, s0 = x.clone(), s.clone()
    _lib.call("lidog_allreduce_f32", _lib.ptr(x), x.numel(), comm)
    _lib.call("lidog_allreduce_f64", _lib.ptr(s), s.numel(), comm)
    torch.cuda.synchronize()
    ok = torch.equal(x, x0) and torch.equal(s, s0)          # one rank: the sum is the identity
    bad = L.lidog_comm_init_rank(uid, 1, 3, ctypes.byref(ctypes.c_void_p()))
    ok = ok and bad != 0 and b"bad rank" in L.lidog_last_error()
    assert L.lidog_comm_destroy(comm) == 0
    q.put(bool(ok))


def test_rccl_collectives_through_the_c_abi():
    """lidog_comm_* / lidog_allreduce_f32 / _f64 (include/lidog_amd.h) against the real RCCL library, one rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_c_abi_worker, args=(q,))
    p.start()
    ok = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0 and ok


def _rccl_executor_worker(q, port):
    """own process, ONE-rank RCCL group, every data-parallel path on: the trunk executor with its collectives issued
    from C on this library's communicators against the operator path -- same bits"""
    import traceback
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        sys.path.insert(0, REPO)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import lidog_amd
        import lidog_amd.me as ME
        from helpers import seeded_state_dict, small_batch
        from lidog_amd import comm, trunk
        from lidog_amd.optim import FlatAdam, GradientBuckets
        from lidog_amd.trainer import LiDOGStep, setup_data_parallel
        ME.MinkowskiSyncBatchNorm.single_rank = GradientBuckets.single_rank = True

        def batch(seeds):
            coords = small_batch(seeds, n_points=2500).cuda()
            g = torch.Generator().manual_seed(seeds[0])
            n = coords.shape[0]
            return {"coords_int": coords, "source_features0": torch.ones((n, 1), device="cuda"),
                    "source_sem_labels0": torch.randint(-1, 7, (n,), generator=g).cuda(),
                    "source_bev_labels0": {"block8": torch.randint(-1, 7, (len(seeds), 17, 17), generator=g).cuda()}}

        runs = {}
        for on in (False, True):
            trunk.set_enabled(on)
            model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=5.0).cuda()
            model.load_state_dict(seeded_state_dict(model, 5))
            model = setup_data_parallel(model).train()
            assert sum(isinstance(m, ME.MinkowskiSyncBatchNorm) for m in model.modules()) == 62
            opt = FlatAdam(model, lr=1e-2, weight_decay=1e-4, bucket_bytes=8 << 20)
            assert opt.buckets.active and opt.buckets.transport.kind == "native" and comm.transport().comm_bn
            step = LiDOGStep(model, opt)
            losses, grads, took = [], [], []
            for it in range(3):
                b = batch((61 + it, 71 + it))
                total, sem_l, bev_l, sem = step.forward_loss(b)
                took.append(type(sem.F.grad_fn).__name__)
                opt.zero_grad()
                total.backward()
                opt.step()
                torch.cuda.synchronize()
                assert opt.strays == 0
                losses.append([float(total.detach()), float(sem_l.detach()), float(bev_l.detach())])
                grads.append(opt.flat.grad.clone())
            assert all((t == "_TrunkFnBackward") == on for t in took), took
            assert opt.buckets.issued_early >= 3
            runs[on] = (losses, grads, {k: v.clone() for k, v in model.state_dict().items()})
        ok, msg = True, ""
        if runs[True][0] != runs[False][0]:
            ok, msg = False, f"losses {runs[True][0]} vs {runs[False][0]}"
        for i, (a, b) in enumerate(zip(runs[True][1], runs[False][1])):
            if not torch.equal(a, b):
                ok, msg = False, msg + f" gradients of step {i} differ by {(a - b).abs().max().item():.3e}"
        for k in runs[True][2]:
            if not torch.equal(runs[True][2][k], runs[False][2][k]):
                ok, msg = False, msg + f" state {k} differs"
                break
        q.put((ok, msg))
        dist.destroy_process_group()
    except Exception as e:
        q.put((False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_executor_with_rccl_collectives_equals_the_operator_path():
    """A data-parallel rank runs the trunk executor: SyncBatchNorm statistics all-reduced on the compute stream
    through lidog_allreduce_f64 from inside lidog_trunk_forward / _backward, gradient buckets reduced from C as their
    last gradient is queued -- against real RCCL (one rank), bit-identical to the operator path's SyncBatchNorm step
    over three optimiser steps (losses, reduced gradients, parameters, running statistics)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_executor_worker, args=(q, 29500 + os.getpid() % 2000))
    p.start()
    ok, msg = q.get(timeout=600)
    p.join(120)
    assert ok, msg
    assert p.exitcode == 0


def _peer_worker(rank, world, port, q):
    import traceback
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LIDOG_PEER_ALLREDUCE="1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path.insert(0, REPO)
        torch.cuda.set_device(0)
        from lidog_amd import _lib, comm
        tr = comm.transport()
        if tr.peer is None:
            q.put((rank, False, "peer all-reduce not available: " + tr.peer_note))
            dist.barrier()
            return
        g = torch.Generator().manual_seed(7)
        a = torch.randn(4096, 4096, device="cuda")
        ok, msg = True, ""
        for it in range(300):
            n = [1, 65, 193, 513, 1026][it % 5]
            # every rank can rebuild every rank's contribution: the expected result is the rank-ordered float64 sum
            parts = [torch.randn(n, generator=torch.Generator().manual_seed(1000 * it + r), dtype=torch.float64) * 10 ** (r - 1)
                     for r in range(world)]
            want = parts[0].clone()
            for r in range(1, world):
                want = want + parts[r]
            t = parts[rank].cuda()
            if it % 7 == 0:
                torch.mm(a, a)            # the kernel also has to get through behind a busy stream
            tr.allreduce_f64(t)
            if not torch.equal(t.cpu(), want):
                ok, msg = False, f"call {it} (n = {n}): max |diff| {(t.cpu() - want).abs().max().item():.3e}"
                break
        if ok and _lib.load().lidog_peer_status(tr.peer) != 0:
            ok, msg = False, "a wait timed out"
        # bigger than the mailbox: falls back to the group's own all-reduce
        big = torch.full((5000,), float(rank + 1), dtype=torch.float64, device="cuda")
        tr.allreduce_f64(big)
        if ok and not torch.equal(big.cpu(), torch.full((5000,), float(sum(range(1, world + 1))), dtype=torch.float64)):
            ok, msg = False, "fallback for large messages"
        q.put((rank, ok, msg))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_peer_allreduce_three_processes_one_gpu():
    """The one-shot peer all-reduce of the SyncBatchNorm statistics messages (csrc/comm.hip: mailboxes opened through
    hipIpc, push + flag + rank-ordered sum) with three processes sharing the one GPU: 300 calls of five message sizes,
    every rank must hold exactly the rank-ordered float64 sum (identical bits on all ranks), no wait may time out.
    What one GPU cannot show is the xGMI memory model between devices: on a multi-GPU node the start-up self-test of
    lidog_amd.comm decides, and RCCL stays the fallback."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 2000
    procs = [ctx.Process(target=_peer_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(3)]
    failed = not all(ok for _, ok, _ in got)
    for p in procs:
        p.join(10 if failed else 60)
        if p.is_alive():
            p.terminate()
            p.join(30)
    assert not failed, got


def test_two_rank_bench_keeps_replicas_identical():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank) with both ranks on the
    one GPU over gloo (LIDOG_BENCH_ONE_GPU=1): the JSON line comes out, the ranks ran the executor with the peer
    all-reduce for the statistics, and after the steps every rank holds bit-identical parameters and SyncBatchNorm
    running statistics (the data-parallel invariant; bench.py checks it over the ranks)"""
    import json
    import subprocess
    env = dict(os.environ, LIDOG_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LIDOG_PEER_ALLREDUCE="1")
    # no launcher environment: `python bench.py --gpus 2` starts its two ranks itself (the driver's torch.distributed.run
    # line, from a parent that never touches the GPU) and relays rank 0's line
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--min-seconds", "0.1", "--batch", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2+syncbn" and line["config"]["global_batch"] == 4
    assert line["config"]["trunk_path"] == "executor" and line["config"]["statistics_allreduce"] == "peer one-shot"
    assert line["replicas_identical"] == {"parameters": True, "syncbn_running_statistics": True}
    assert line["rccl_ranks_seen"]["torch_process_group"] == 2 and line["config"]["peer_note"] == "on"
    assert "peer_one_shot" in line["statistics_allreduce_us"] and "peer_error" not in line
    # phase 2 (bench.launch_ranks -> probe_phase): AFTER the line, a second group of fresh processes measured one
    # statistics message through the peer one-shot path (probe mode) and torch.distributed; stderr carries its JSON line
    probe = [l for l in out.stderr.splitlines() if l.startswith('{"peer_probe"')]
    assert probe, out.stderr[-3000:]
    pr = json.loads(probe[-1])["peer_probe"]
    assert pr["ranks"] == 2 and pr["one_gpu"] and pr["peer_sum_ok"] is True and pr["peer_error"] is None
    assert pr["us"]["peer_one_shot"] > 0 and pr["us"]["torch_distributed"] > 0
    assert "its exit code is ignored" in out.stderr


def test_ranks_started_by_a_foreign_launcher_run_the_probe_themselves():
    """the driver starts the ranks with its own `python -m torch.distributed.run ... bench.py --gpus N` line: no parent of
    this repository exists, so every rank starts its probe child after it has printed, torn down and left the group"""
    import json
    import socket
    import subprocess
    env = dict(os.environ, LIDOG_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LIDOG_PEER_ALLREDUCE"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--min-seconds", "0.1", "--batch", "1", "--config", "source8k"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["statistics_allreduce"] != "peer one-shot"   # the step itself: default path
    probe = [l for l in out.stderr.splitlines() if '{"peer_probe"' in l]
    assert probe, out.stderr[-3000:]
    pr = json.loads(probe[-1][probe[-1].index('{"peer_probe"'):])["peer_probe"]
    assert pr["ranks"] == 2 and pr["peer_sum_ok"] is True and pr["us"]["peer_one_shot"] > 0


def test_a_hung_two_rank_run_still_produces_a_fallback_line():
    """rank 1 never issues its first gradient bucket (LIDOG_BENCH_FAULT), rank 0 waits in that
    all-reduce for ever.  Launched the way the driver launches N > 1 (its own torch.distributed.run line; both ranks on the
    one GPU over gloo): every rank's supervisor (bench.supervise_rank) ends its silent worker, a fresh pair of workers runs
    the measurement in safe mode, rank 0 prints THAT line marked "safe-fallback", and the command exits non-zero."""
    import json
    import socket
    import subprocess
    env = dict(os.environ, LIDOG_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LIDOG_BENCH_FAULT="bucket_hang:1",
               LIDOG_BENCH_WATCHDOG_S="40")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LIDOG_PEER_ALLREDUCE", "LIDOG_DP_SAFE"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--min-seconds", "0.1", "--batch", "1", "--config", "source8k"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode != 0
    assert "injected fault" in out.stderr and "was killed" in out.stderr, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-3000:])
    line = lines[0]
    assert line["mode"] == "safe-fallback" and line["n_gpus"] == 2 and "dp_safe" in line["config"]
    assert line["config"]["collectives"] == "torch" and line["config"]["trunk_path"] == "executor"
    assert line["replicas_identical"] == {"parameters": True, "syncbn_running_statistics": True}
