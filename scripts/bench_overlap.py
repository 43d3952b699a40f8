"""How much of the per-row reduction could hide behind the gathered GEMM?  Runs the GEMM of one layer and the reduction
of another product-row buffer (same size) on two streams at once and compares with running them back to back."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth
from lidog_amd._lib import call, ptr
b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
s2 = torch.cuda.Stream()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("%-16s | %8s %8s %8s | %8s %8s" % ("layer", "gemm", "reduce", "serial", "overlap", "saved"))
for s, Cin, Cout in [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 256, 256), (16, 256, 256)]:
    m = cm.kernel_map(s, s, 3)
    x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda"); T2 = torch.randn(m.P, Cout, device="cuda"); out = torch.empty(m.n_out, Cout, device="cuda")
    gemm = lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    red = lambda: call("lidog_sconv_reduce", ptr(T2), ptr(m.pos_out), m.n_out, m.K, Cout, None, None, ptr(out))
    def both():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            red()
        gemm()
        main.wait_stream(s2)
    def serial():
        gemm(); red()
    tg, tr, ts, tb = timeit(gemm), timeit(red), timeit(serial), timeit(both)
    print("s%-2d %3d->%3d      | %8.3f %8.3f %8.3f | %8.3f %7.1f%%" % (s, Cin, Cout, tg, tr, ts, tb, 100 * (ts - tb) / ts))
