"""cProfile of the host side of training steps: CONFIG=kitti120k|source8k python scripts/host_profile.py [steps]"""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.train import build_model, build_step
cfg = os.environ.get("CONFIG", "kitti120k")
kind = "MinkUNet34BEV" if cfg == "kitti120k" else "MinkUNet34"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(1234)
model, step, _ = build_step(build_model(kind), kind)
batches = [synth.make_batch(range(4 * i, 4 * i + 4), cfg, "cuda") for i in range(2)]
ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
for i in range(4):
    step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(steps):
    step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
pr.disable()
torch.cuda.synchronize()
print(f"{cfg}: {steps} steps")
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
