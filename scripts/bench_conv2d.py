"""Micro-benchmark of the BEV head convolutions at bench shapes (bs 4)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from lidog_amd._lib import call, ptr
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = int(os.environ.get("BS", 4))
for Cin, H, Cout in ((96, 666, 256), (256, 333, 256)):
    Ho = (H + 2 - 3) // 2 + 1
    x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, Cout, Ho, Ho, device="cuda"); gy = torch.randn_like(y); gx = torch.empty_like(x); gw = torch.empty_like(w)
    ws = torch.empty(9 * Cin * Cout, device="cuda"); ws2 = torch.empty(32 * w.numel(), device="cuda")
    fl = 2.0 * B * Ho * Ho * Cout * Cin * 9 / 1e9
    tf = timeit(lambda: call("lidog_conv2d_fwd", ptr(x), ptr(w), None, B, Cin, H, H, Cout, 3, 2, 1, ptr(y)))
    td = timeit(lambda: call("lidog_conv2d_dgrad", ptr(gy), ptr(w), B, Cin, H, H, Cout, 3, 2, 1, ptr(gx), ptr(ws)))
    tw = timeit(lambda: call("lidog_conv2d_wgrad", ptr(x), ptr(gy), B, Cin, H, H, Cout, 3, 2, 1, ptr(gw), None, ptr(ws2), ws2.numel()))
    print(f"conv {Cin}->{Cout} {H}x{H}: fwd {tf:.3f} ms {fl/tf:.1f} TF | dgrad {td:.3f} ms {fl/td:.1f} TF | wgrad {tw:.3f} ms {fl/tw:.1f} TF")
