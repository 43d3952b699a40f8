"""training steps only (no eval / cpu baseline): the command profiled for profiles/*kernel_stats*"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, LiDOGStep, SourceStep
torch.manual_seed(1234)
cfg = os.environ.get("CONFIG", "kitti120k")     # source8k: BASELINE config 1 (MinkUNet34, SoftDICE only)
if cfg == "kitti120k":
    model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
    step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
else:
    model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    step = SourceStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
batches = [synth.make_batch(range(4 * i, 4 * i + 4), cfg, "cuda") for i in range(2)]
READY = torch.cuda.Event(); READY.record(); torch.cuda.synchronize()
n = int(os.environ.get("STEPS", 5))
for i in range(n):
    step.training_step(batches[i % 2], prefetch=None if os.environ.get('NO_PREFETCH') else batches[(i + 1) % 2], prefetch_ready=READY)
torch.cuda.synchronize()
print("steps", n)
