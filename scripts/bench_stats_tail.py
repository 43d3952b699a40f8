"""Latency of the statistics reductions on SMALL inputs (the in-kernel finish dominates): lidog_bn_stats and
lidog_bn_bwd_reduce over [n, C] for the deep layers' shapes, per call, back to back on one stream."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidog_amd import _lib
L = _lib.load()
ptr = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
for n, C in ((4_000, 256), (15_000, 256), (41_000, 256), (102_000, 128), (208_000, 96), (352_000, 96), (352_000, 32)):
    x = torch.randn(n, C, device="cuda")
    dy = torch.randn(n, C, device="cuda")
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device="cuda")
    mean = torch.zeros(C, device="cuda"); invstd = torch.ones(C, device="cuda")
    ws = torch.empty(int(L.lidog_bn_reduce_ws(C, 1)), dtype=torch.float64, device="cuda")
    dw = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    res = []
    for name, fn in (("stats", lambda: _lib.call("lidog_bn_stats", ptr(x), n, C, 1, ptr(sums), ptr(ws), float(n), 1e-5, 0.1, ptr(mean), ptr(invstd), None, None)),
                     ("bwd_reduce", lambda: _lib.call("lidog_bn_bwd_reduce", ptr(dy), ptr(x), None, n, C, 1, ptr(mean), ptr(invstd), ptr(sums), ptr(ws), float(n), ptr(dw), ptr(db), None, None))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append("%s %6.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 200))
    print("n %7d C %3d: %s   (bytes/%s)" % (n, C, " | ".join(res), "call %.1f MB" % (n * C * 4 / 1e6)))
