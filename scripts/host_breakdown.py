"""host enqueue time of one training step by phase (no synchronisation inside the step): CONFIG=kitti120k|source8k"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, lidog_amd
from lidog_amd import synth
from lidog_amd.train import build_model, build_step
cfg = os.environ.get("CONFIG", "kitti120k")
kind = "MinkUNet34BEV" if cfg == "kitti120k" else "MinkUNet34"
torch.manual_seed(1234)
model, step, _ = build_step(build_model(kind), kind)
batches = [synth.make_batch(range(4 * i, 4 * i + 4), cfg, "cuda") for i in range(2)]
READY = torch.cuda.Event(); READY.record(); torch.cuda.synchronize()
def one(i, timed):
    b, nxt = batches[i % 2], batches[(i + 1) % 2]
    t = [time.perf_counter()]
    if kind == "MinkUNet34BEV":
        total, *_ = step.forward_loss(b)
    else:
        st = step._sparse_input(b)
        total = step.criterion(step.model(st, is_seg=True).F, b["source_sem_labels0"].long())
    t.append(time.perf_counter())
    step.opt.zero_grad()
    total.backward()
    t.append(time.perf_counter())
    step.opt.step()
    t.append(time.perf_counter())
    step._after_step(nxt, READY)
    t.append(time.perf_counter())
    if timed:
        torch.cuda.synchronize()
        t.append(time.perf_counter())
    return t
for i in range(4):
    one(i, False)
torch.cuda.synchronize()
acc = [0.0] * 5
N = 8
for i in range(N):
    t = one(i, True)
    for j in range(5):
        acc[j] += (t[j + 1] - t[j]) * 1e3 / N
print(f"{cfg}: forward+loss {acc[0]:.2f} ms, backward {acc[1]:.2f}, optimiser {acc[2]:.2f}, prepare next maps {acc[3]:.2f}, "
      f"GPU tail after enqueue {acc[4]:.2f}  | enqueue total {sum(acc[:4]):.2f} ms")
