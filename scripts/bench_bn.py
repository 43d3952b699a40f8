"""Micro-benchmark of the BatchNorm passes at the bench workload's [rows, C] shapes (bs 4)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from lidog_amd._lib import call, ptr
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0] * 4
for n, C, cnt in ((304000, 96, 5), (304000, 32, 1), (179000, 96, 5), (179000, 32, 5), (89000, 128, 5), (89000, 64, 7), (37500, 256, 5),
                  (37500, 128, 9), (14000, 256, 13)):
    x = torch.randn(n, C, device="cuda"); dy = torch.randn(n, C, device="cuda"); y = torch.randn(n, C, device="cuda")
    res = torch.randn(n, C, device="cuda"); out = torch.empty_like(x); dres = torch.empty_like(x)
    sums = torch.zeros(2 * C + 1, dtype=torch.float64, device="cuda")
    mean = torch.zeros(C, device="cuda"); inv = torch.ones(C, device="cuda"); w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ws = torch.empty(512 * 2 * C, dtype=torch.float64, device="cuda")
    dw = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    by = 4.0 * n * C / 1e6  # MB per pass
    t_s = timeit(lambda: call("lidog_bn_stats", ptr(x), n, C, 1, ptr(sums), ptr(ws), float(n), 0.0, 0.0, None, None, None, None))
    t_a = timeit(lambda: call("lidog_bn_apply", ptr(x), n, C, 1, ptr(mean), ptr(inv), ptr(w), ptr(b), ptr(res), 1, ptr(out)))
    t_r = timeit(lambda: call("lidog_bn_bwd_reduce", ptr(dy), ptr(x), ptr(y), n, C, 1, ptr(mean), ptr(inv), ptr(sums), ptr(ws), float(n), None, None, None, None))
    t_b = timeit(lambda: call("lidog_bn_bwd_apply", ptr(dy), ptr(x), ptr(y), n, C, 1, ptr(mean), ptr(inv), ptr(w), ptr(sums), float(n), ptr(out), ptr(dres), ptr(dw), ptr(db), None))
    print(f"n={n:7d} C={C:3d}: stats {1e3*t_s:6.1f} us {by/t_s/1e3:5.2f} TB/s | apply {1e3*t_a:6.1f} us {3*by/t_a/1e3:5.2f} TB/s | "
          f"bwd_reduce {1e3*t_r:6.1f} us {3*by/t_r/1e3:5.2f} TB/s | bwd_apply {1e3*t_b:6.1f} us {5*by/t_b/1e3:5.2f} TB/s")
    for i, t in enumerate((t_s, t_a, t_r, t_b)): tot[i] += t * cnt
print("weighted ms/step: stats %.2f apply %.2f bwd_reduce %.2f bwd_apply %.2f" % tuple(tot))
