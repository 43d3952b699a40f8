"""host cost and blocking behaviour of the statistics all-reduce on a one-rank RCCL group:
native (lidog_allreduce_f64 on the compute stream) vs torch.distributed.all_reduce, with the GPU busy"""
import os, socket, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.distributed as dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from lidog_amd import _lib, comm
tr = comm.transport()
print("transport", tr.kind)
a = torch.randn(8192, 8192, device="cuda")
s = torch.zeros(193, dtype=torch.float64, device="cuda")
def busy():
    for _ in range(20):
        torch.mm(a, a)
for name, fn in (("native", lambda: tr.allreduce_f64(s)), ("torch", lambda: dist.all_reduce(s))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: idle GPU  {1e6*(t1-t0)/200:.1f} us per call on the host, drain {1e3*(t2-t1):.2f} ms")
    busy(); t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: busy GPU  {1e6*(t1-t0)/200:.1f} us per call on the host, drain {1e3*(t2-t1):.2f} ms (blocking if the first figure ~ GPU time / 200)")
dist.destroy_process_group()
