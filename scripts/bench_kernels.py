"""Micro-benchmark of the sparse-conv kernels on the bench workload's real kernel maps (bs 4, kitti120k)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth
from lidog_amd._lib import call, ptr

bs = int(os.environ.get("BS", 4))
b = synth.make_batch(range(bs), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

from lidog_amd import _lib
L = _lib.load()
L.lidog_set_sparse_core(int(os.environ.get("CORE", 1)))
print("sparse core:", L.lidog_get_sparse_core())
cases = [(1, 3, 96, 96), (1, 3, 128, 96), (2, 3, 96, 96), (2, 3, 32, 32), (4, 3, 128, 128), (4, 3, 64, 64), (8, 3, 256, 256), (8, 3, 128, 128),
         (8, 3, 384, 256), (16, 3, 256, 256), (4, 3, 192, 128)]
print("%-22s %9s | %8s %7s | %8s %8s | %8s %7s | %8s %7s" % ("layer", "P", "gemm ms", "TF/s", "red(tbl)", "red(rows)", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
tot = [0, 0, 0, 0]
for s, k, Cin, Cout in cases:
    m = cm.kernel_map(s, s, k)
    x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    g = torch.randn(m.n_out, Cout, device="cuda"); Wt = W.transpose(1, 2).contiguous()
    T = torch.empty(m.P, Cout, device="cuda"); out = torch.empty(m.n_out, Cout, device="cuda")
    T2 = torch.empty(m.P, Cin, device="cuda"); gx = torch.empty(m.n_in, Cin, device="cuda"); gW = torch.empty_like(W)
    items, ns, item_off = ME._wgrad_items(m, Cin, Cout)
    slabs = L.lidog_sconv_wgrad_slabs(Cin, Cout, ns)
    part = torch.empty(max(slabs, 1), Cin, Cout, device="cuda")
    t_g = timeit(lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None))
    t_r0 = timeit(lambda: call("lidog_sconv_reduce", ptr(T), ptr(m.pos_out), m.n_out, m.K, Cout, None, None, ptr(out)))
    rp, rl = m.rows("out")
    out2 = torch.empty_like(out)
    t_r = timeit(lambda: call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), m.n_out, Cout, None, None, ptr(out2)))
    assert torch.equal(out, out2), "row-list reduction differs from the table walk"
    t_d = timeit(lambda: ME._gemm(g, m.pair_out, Wt, None, m, Cout, Cin, T2, None))
    t_w = timeit(lambda: call("lidog_sconv_wgrad", ptr(x), ptr(m.pair_in), ptr(g), ptr(m.pair_out), ptr(items), ns, ptr(item_off), m.K, Cin, Cout, ptr(part), ptr(gW)))
    fl = 2.0 * m.P * Cin * Cout / 1e9
    print("s%-2d k%d %3d->%3d ns=%-3d %9d | %8.3f %7.1f | %8.3f %8.3f | %8.3f %7.1f | %8.3f %7.1f" % (s, k, Cin, Cout, ns, m.P, t_g, fl / t_g, t_r0, t_r, t_d, fl / t_d, t_w, fl / t_w))
    for i, t in enumerate((t_g, t_r, t_d, t_w)): tot[i] += t
print("sum ms: gemm %.2f reduce %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
