"""host enqueue time vs GPU time of one training step (is python the bottleneck?)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, LiDOGStep
torch.manual_seed(1234)
model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
batches = [synth.make_batch(range(4 * i, 4 * i + 4), "kitti120k", "cuda") for i in range(2)]
READY = torch.cuda.Event(); READY.record(); torch.cuda.synchronize()
for i in range(3):
    step.training_step(batches[i % 2], prefetch=None if os.environ.get('NO_PREFETCH') else batches[(i + 1) % 2], prefetch_ready=READY)
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    step.training_step(batches[i % 2], prefetch=None if os.environ.get('NO_PREFETCH') else batches[(i + 1) % 2], prefetch_ready=READY)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
if os.environ.get("CPROF"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(4):
        step.training_step(batches[i % 2], prefetch=None if os.environ.get('NO_PREFETCH') else batches[(i + 1) % 2], prefetch_ready=READY)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
