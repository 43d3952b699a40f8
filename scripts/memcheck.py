"""Does a long run leak device memory?  N training steps over 6 different batches (different voxel counts: the executor's
arenas and the allocator's pools have to settle), memory_allocated / memory_reserved printed every 40 steps."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from lidog_amd import synth
from lidog_amd.train import build_model, build_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
torch.manual_seed(0)
model, step, _ = build_step(build_model("MinkUNet34BEV"), "MinkUNet34BEV")
batches = [synth.make_batch(range(4 * i, 4 * i + 4), "kitti120k", "cuda") for i in range(6)]
ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
for i in range(steps):
    out = step.training_step(batches[i % 6], prefetch=batches[(i + 1) % 6], prefetch_ready=ready)
    if i % 40 == 39:
        torch.cuda.synchronize()
        print(f"step {i + 1}: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB, loss {float(out['loss']):.4f}", flush=True)
