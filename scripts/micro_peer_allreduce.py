"""Latency of one SyncBatchNorm statistics message (MinkowskiSyncBatchNorm, train_lidog.py:228) through every transport
this library has: the one-shot peer all-reduce (csrc/comm.hip), this library's RCCL communicator, torch.distributed.

Two ways to run it:

* under a launcher (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment: `python -m torch.distributed.run
  --nproc-per-node N scripts/micro_peer_allreduce.py`, which is what bench.py's second phase does after the headline
  line has been printed): one rank per GPU over RCCL (LIDOG_BENCH_ONE_GPU=1: every rank on cuda:0 over gloo), the peer
  path set up in `probe` mode (self-tested, measured, never used by a training step).  Rank 0 prints ONE JSON line
  {"peer_probe": ...} to stderr and writes it to gpurun_out/peer_probe_n<N>.json.  Every wait is bounded; the caller
  gives the whole group a watchdog and ignores its exit code.
* stand-alone `python scripts/micro_peer_allreduce.py [N]`: N processes sharing the one GPU (no xGMI hop: the kernel's
  own cost), message sizes 65 / 193 / 513 / 1026 doubles.
"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _time_us(fn, msg, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        msg.fill_(1.0)
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / reps, 2)


def launched():
    """one rank of the probe group (bench.py phase 2)"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one_gpu = os.environ.get("LIDOG_BENCH_ONE_GPU") == "1"
    local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", 0))
    os.environ["LIDOG_PEER_ALLREDUCE"] = "probe"
    os.environ.setdefault("LIDOG_PEER_SPIN_LIMIT", str(1 << 22))     # a few seconds per wait at most, not minutes
    torch.cuda.set_device(local)
    if one_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from lidog_amd import _lib as L, comm
    t0 = time.time()
    tr = comm.transport()
    setup_s = time.time() - t0
    msg = torch.ones(193, dtype=torch.float64, device="cuda")      # one 96-channel message: (sum, sum, rows)
    out = {"ranks": world, "backend": dist.get_backend(), "devices_seen": torch.cuda.device_count(),
           "one_gpu": one_gpu, "peer_note": tr.peer_note, "setup_s": round(setup_s, 2), "message_doubles": 193, "us": {}}
    pc = tr.peer or tr.peer_probe
    if pc:
        def _peer():
            tr.peer_bind(pc)
            L.call("lidog_peer_allreduce_f64", pc, L.ptr(msg), msg.numel())
        out["us"]["peer_one_shot"] = _time_us(_peer, msg, 300)
        # the sums are right on every rank after the timed calls too
        msg.fill_(float(rank + 1))
        _peer()
        torch.cuda.synchronize()
        out["peer_sum_ok"] = bool((msg == world * (world + 1) / 2).all().item())
    if tr.comm_bn:
        out["us"]["rccl"] = _time_us(lambda: L.call("lidog_allreduce_f64", L.ptr(msg), msg.numel(), tr.comm_bn), msg, 300)
    out["us"]["torch_distributed"] = _time_us(lambda: dist.all_reduce(msg), msg, 300 if dist.get_backend() == "nccl" else 5)
    try:
        tr.check()
        out["peer_error"] = None
    except RuntimeError as e:
        out["peer_error"] = str(e)
    # what 241 messages per step would cost on the dependent chain
    out["ms_per_step_241_messages"] = {k: round(241 * v / 1e3, 3) for k, v in out["us"].items()}
    dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        line = json.dumps({"peer_probe": out})
        sys.stderr.write(line + "\n")
        sys.stderr.flush()
        try:
            d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", R), "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, f"peer_probe_n{world}.json"), "w") as f:
                f.write(line + "\n")
        except OSError:
            pass
    comm.reset()
    dist.destroy_process_group()


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LIDOG_PEER_ALLREDUCE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lidog_amd import comm
    tr = comm.transport()
    assert tr.peer is not None, tr.peer_note
    for n in (65, 193, 513, 1026):
        t = torch.ones(n, dtype=torch.float64, device="cuda")
        for _ in range(20):
            tr.allreduce_f64(t)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            tr.allreduce_f64(t)
        e1.record()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"world {world} n {n}: {1e3 * e0.elapsed_time(e1) / 500:.1f} us per all-reduce (back to back on one stream)")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        launched()
    else:
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        mp.spawn(worker, args=(world, 29650), nprocs=world)
