"""training steps only (no eval / cpu baseline): the command profiled for profiles/*kernel_stats*"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, LiDOGStep, SourceStep
torch.manual_seed(1234)
cfg = os.environ.get("CONFIG", "kitti120k")     # source8k: BASELINE config 1 (MinkUNet34, SoftDICE only)
DP = os.environ.get("DP") == "1"                # one-rank RCCL group with every data-parallel path on (as bench.py's
if DP:                                          # LIDOG_BENCH_SINGLE_RANK_DP=1)
    import socket
    import torch.distributed as dist
    import lidog_amd.me as ME
    from lidog_amd.trainer import GradientBuckets, setup_data_parallel
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ME.MinkowskiSyncBatchNorm.single_rank = GradientBuckets.single_rank = True
BS = int(os.environ.get("BS", 4))               # scans per step (config 5, highres524k: BS=1)
if cfg in ("kitti120k", "highres524k"):
    model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
    if DP:
        model = setup_data_parallel(model)
    step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
else:
    model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    step = SourceStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
batches = [synth.make_batch(range(BS * i, BS * i + BS), cfg, "cuda") for i in range(2)]
READY = torch.cuda.Event(); READY.record(); torch.cuda.synchronize()
n = int(os.environ.get("STEPS", 5))
import contextlib
prio = os.environ.get("LIDOG_MAIN_STREAM_PRIORITY", "-1" if DP else "none")     # as bench.py
ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(prio))) if prio != "none" else contextlib.nullcontext()
with ctx:
    for i in range(n):
        step.training_step(batches[i % 2], prefetch=None if os.environ.get('NO_PREFETCH') else batches[(i + 1) % 2], prefetch_ready=READY)
torch.cuda.synchronize()
print("steps", n, getattr(step, "last_path", ""))
if DP:
    dist.destroy_process_group()
