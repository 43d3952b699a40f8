"""Does the step wait for its coordinate maps?  Times, with events on the step's stream, the wait in
CoordinateManager.handover (the forward pass of step i waits for the maps prepared during step i - 1 on the side
stream) over 30 training steps of the bench workload, and the step time with the maps of both batches built once and
kept (LIDOG_EXP_REUSE_MAPS=1: no map kernels at all in the timed steps)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import lidog_amd, lidog_amd.me as ME
from lidog_amd import synth
from lidog_amd.train import build_model, build_step

torch.manual_seed(1234)
model, step, _ = build_step(build_model("MinkUNet34BEV", bound_2d=50.0), "MinkUNet34BEV", optimizer="Adam", lr=1e-3, weight_decay=1e-4)
batches = [synth.make_batch(range(4 * i, 4 * i + 4), "kitti120k", "cuda") for i in range(2)]
ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
waits = []
orig = ME.CoordinateManager.handover
def handover(self):
    if self._ready is None:
        return orig(self)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(self)
    e1.record()
    waits.append((e0, e1))
ME.CoordinateManager.handover = handover
for i in range(8):
    step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
torch.cuda.synchronize(); waits.clear()
t0 = time.perf_counter()
n = 30
for i in range(n):
    step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
w = [a.elapsed_time(b) for a, b in waits]
print(f"step {1e3 * dt:.2f} ms; wait for the prepared maps on the step's stream: mean {sum(w) / len(w):.3f} ms, max {max(w):.3f} ms over {len(w)} steps")
