"""Secondary workloads of BASELINE.json (not the headline bench): training-step throughput of
  C1  train_source.py  MinkUNet34, 8 k-point scans, 0.1 m voxels, bs 4          (configs[0])
  C4  train_aug_based.py  MinkUNet34 on Mix3D nuScenes-like unions of two scans, bs 4   (configs[3])
on one GPU:  python scripts/bench_configs.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import lidog_amd
from lidog_amd import synth
from lidog_amd.trainer import FlatAdam, SourceStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, cfg, mix in (("C1 source8k", "source8k", False), ("C4 mix3d nusc35k", "nusc35k", True)):
    torch.manual_seed(0)
    model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    step = SourceStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
    batches = [synth.make_batch(range(4 * i, 4 * i + 4), cfg, "cuda", mix3d=mix) for i in range(2)]
    nvox = sum(b["coords_int"].shape[0] for b in batches) / 8
    ready = torch.cuda.Event(); ready.record(); torch.cuda.synchronize()
    for i in range(3):
        step.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step.training_step(batches[(i + 1) % 2], prefetch=batches[i % 2], prefetch_ready=ready)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {nvox:.0f} voxels/scan, bs 4: {4 * steps / dt:.1f} scans/s, {1e3 * dt / steps:.2f} ms/step, loss {float(out['loss']):.4f}")
