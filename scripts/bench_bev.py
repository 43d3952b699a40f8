"""Micro-benchmark of sparse2super (winner map + fused view-scramble max-pool) at bench shape (bs 4)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, bev
b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=torch.randn(b["coords_int"].shape[0], 96, device="cuda").relu_().requires_grad_())
def run():
    out = bev.sparse2super(st, bound=50.0, voxel=0.05, pool=(5, 3, 1))
    return out
for tag, fn in (("fwd", run),):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): o = fn()
    e1.record(); torch.cuda.synchronize()
    print(tag, e0.elapsed_time(e1) / 5, "ms", tuple(o.shape), "nonzero frac", float((o != 0).float().mean()))
g = torch.randn_like(o)
e0.record()
for _ in range(5):
    o = run(); o.backward(g)
e1.record(); torch.cuda.synchronize()
print("fwd+bwd", e0.elapsed_time(e1) / 5, "ms")
# ---- kernel-level isolation
from lidog_amd._lib import call, ptr
feats = st.F.detach().contiguous(); n, C = feats.shape
lx, ly, lo, H = bev._device_luts(50.0, 0.05, feats.device); W = H
Ho = Wo = (H + 2 - 5) // 3 + 1
winner = torch.full((4, H, W), -1, dtype=torch.int32, device="cuda"); pixel = torch.empty(n, dtype=torch.int32, device="cuda")
empty = winner.clone()
call("lidog_bev_winner", ptr(st.C), n, ptr(lx), ptr(ly), lo, lx.shape[0], H, W, ptr(winner), ptr(pixel))
out = torch.empty((4, C, Ho, Wo), device="cuda"); arg = torch.empty((4, C, Ho, Wo), dtype=torch.int32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
print("pool kernel        ", t(lambda: call("lidog_bev_pool_fwd", ptr(feats), C, ptr(winner), ptr(pixel), n, 4, H, W, 5, 3, 1, Ho, Wo, ptr(out), ptr(arg), None)), "ms")
print("pool kernel (empty)", t(lambda: call("lidog_bev_pool_fwd", ptr(feats), C, ptr(empty), ptr(pixel), n, 4, H, W, 5, 3, 1, Ho, Wo, ptr(out), ptr(arg), None)), "ms")
print("fill out+arg       ", t(lambda: (out.fill_(0), arg.fill_(-1))), "ms")
print("occupied pixels", int((winner >= 0).sum()), "of", winner.numel())
