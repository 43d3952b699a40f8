"""Output-stationary convolution (csrc/sconv_os.hip) against the two-pass path on the bench workload's kernel maps:
bit-equality and time."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr
b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
L = _lib.load()
BR = L.lidog_sconv_os_block_rows()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 128, 128), (2, 64, 32), (4, 32, 64)]
print("%-16s %9s | %8s %8s %8s | %8s %7s | %s" % ("layer", "P", "gemm", "reduce", "2-pass", "os ms", "x", "equal fwd / dgrad"))
for s, Cin, Cout in cases:
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    x = torch.randn(n, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda"); ref = torch.empty(n, Cout, device="cuda")
    nb = (n + BR - 1) // BR
    seg = torch.empty((m.K, nb + 1), dtype=torch.int32, device="cuda")
    call("lidog_sconv_os_segments", ptr(m.pair_out), ptr(m.k_off), m.K, n, ptr(seg))
    out = torch.empty(n, Cout, device="cuda")
    part = torch.empty(nb * 2 * Cout, dtype=torch.float64, device="cuda")
    t_g = timeit(lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None))
    t_r = timeit(lambda: call("lidog_sconv_reduce", ptr(T), ptr(m.pos_out), n, m.K, Cout, None, None, ptr(ref)))
    t_o = timeit(lambda: call("lidog_sconv_os", ptr(x), ptr(m.pair_in), ptr(m.pair_out), ptr(seg), m.K, n, ptr(W), 0, Cin, Cout, ptr(out), ptr(part)))
    eq = torch.equal(out, ref)
    # data gradient: A = g [n, Cout], Wt [K][Cout][Cin]
    g = torch.randn(n, Cout, device="cuda"); Wt = W.transpose(1, 2).contiguous()
    T2 = torch.empty(m.P, Cin, device="cuda"); gref = torch.empty(n, Cin, device="cuda"); gx = torch.empty(n, Cin, device="cuda")
    ME._gemm(g, m.pair_out, Wt, None, m, Cout, Cin, T2, None)
    call("lidog_sconv_reduce", ptr(T2), ptr(m.pos_in), n, m.K, Cin, None, None, ptr(gref))
    ok2 = "-"
    if Cout <= 128:
        call("lidog_sconv_os", ptr(g), ptr(m.pair_in), ptr(m.pair_out), ptr(seg), m.K, n, ptr(Wt), 1, Cout, Cin, ptr(gx), None)
        ok2 = torch.equal(gx, gref)
    # BN partial sums against a direct fp64 sum
    sums = part.view(nb, 2 * Cout).sum(0)
    d = out.double()
    sok = torch.allclose(sums[:Cout], d.sum(0), rtol=1e-12, atol=1e-9) and torch.allclose(sums[Cout:], (d * d).sum(0), rtol=1e-12, atol=1e-9)
    print("s%-2d %3d->%3d      %9d | %8.3f %8.3f %8.3f | %8.3f %6.2fx | %s / %s  stats %s" % (s, Cin, Cout, m.P, t_g, t_r, t_g + t_r, t_o, (t_g + t_r) / t_o, eq, ok2, sok))
