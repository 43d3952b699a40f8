"""GEMM followed by the reduction (the real sequence: the reduction reads product rows the GEMM has just written)"""
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
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for s, Cin, Cout in [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 256, 256)]:
    m = cm.kernel_map(s, s, 3)
    x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda"); out = torch.empty(m.n_out, Cout, device="cuda")
    rp, rl = m.rows("out")
    def pair():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
        call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), m.n_out, Cout, None, None, ptr(out))
    t_old = timeit(pair)
    T2 = torch.empty(m.P, Cout, device="cuda"); out2 = torch.empty(m.n_out, Cout, device="cuda")
    rpm, rlm = m.rows("out", True)
    nc = m.tiles_nc
    def gemm_nc():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T2, None, nc)
    def cred():
        call("lidog_sconv_center_reduce", ptr(x), ptr(W[m.center]), ptr(T2), ptr(rpm), ptr(rlm), m.n_out, Cin, Cout, None, None,
             ptr(out2), None, None, float(m.n_out), 0.0, 0.0, None, None, None, None)
    def pair2():
        gemm_nc(); cred()
    t_new = timeit(pair2)
    print("s%-2d %3d->%3d  gemm+reduce %.3f ms | without centre: gemm %.3f + centre-reduce %.3f = %.3f ms  (%.2fx)  equal %s"
          % (s, Cin, Cout, t_old, timeit(gemm_nc), timeit(cred), t_new, t_old / t_new, torch.equal(out, out2)))
