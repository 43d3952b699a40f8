"""latency of the one-shot peer all-reduce, N processes sharing the one GPU (no xGMI hop: the kernel's own cost)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.distributed as dist, torch.multiprocessing as mp


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
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            tr.allreduce_f64(t)
        e1.record(); torch.cuda.synchronize()
        if rank == 0:
            print(f"world {world} n {n}: {1e3 * e0.elapsed_time(e1) / 500:.1f} us per all-reduce (back to back on one stream)")
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29650), nprocs=world)
