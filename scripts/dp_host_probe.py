"""one-rank data-parallel step: host issue time per step vs wall time per step (is the step host-bound?)"""
import os, socket, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch, torch.distributed as dist, lidog_amd
import lidog_amd.me as ME
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, LiDOGStep, GradientBuckets, setup_data_parallel
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ME.MinkowskiSyncBatchNorm.single_rank = GradientBuckets.single_rank = True
torch.manual_seed(1234)
model = setup_data_parallel(lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train())
step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
batches = [synth.make_batch(range(4 * i, 4 * i + 4), "kitti120k", "cuda") for i in range(2)]
READY = torch.cuda.Event(); READY.record(); torch.cuda.synchronize()
prio = os.environ.get("LIDOG_MAIN_STREAM_PRIORITY", "-1")
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(prio))) if prio != "none" else contextlib.nullcontext()
with ctx:
    for i in range(6):
        step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=READY)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter(); issue = 0.0
    for i in range(n):
        a = time.perf_counter()
        step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=READY)
        issue += time.perf_counter() - a
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
from lidog_amd.comm import transport
print(f"transport {transport().kind}/{transport().bucket_kind} prio {prio} path {step.last_path}: host issue {1e3*issue/n:.1f} ms/step, "
      f"loop {1e3*(t1-t0)/n:.1f}, with drain {1e3*(t2-t0)/n:.1f} ms/step")
dist.destroy_process_group()
