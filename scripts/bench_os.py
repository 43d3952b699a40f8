"""Output-stationary 3^3 convolution (csrc/sconv_os.hip) against the two-pass path (gathered GEMM -> product rows ->
per-row reduction) on the bench workload's kernel maps: results must be torch.equal, forward and data gradient; times of
both.    BS=4 python scripts/bench_os.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr

bs = int(os.environ.get("BS", 4))
b = synth.make_batch(range(bs), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
L = _lib.load()


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sorted_rows(m):
    n = m.n_out
    pad = (n + 127) // 128 * 128
    perm = torch.empty(pad, dtype=torch.int32, device="cuda")
    wm = torch.empty(pad // 32, dtype=torch.int32, device="cuda")
    order = torch.empty(pad // 128, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: call("lidog_kernel_map_sorted", ptr(m.nbr), n, m.K, ptr(m.k_off), ptr(perm), ptr(wm), ptr(order), ptr(ws), ws.numel()), reps=5)
    return perm, wm, order, t


CHECK = os.environ.get("NOCHECK") != "1"
cases = [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 128, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (4, 192, 128), (8, 256, 256),
         (8, 128, 128), (8, 384, 256), (16, 256, 256)]
if os.environ.get("CASES"):
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["CASES"].split(";")]
print("%-16s %9s %6s | %8s %8s %8s | %8s %7s %6s | %8s %8s %6s" % ("layer", "P", "fill32", "gemm", "reduce", "two-pass", "os fwd", "TF/s", "ratio", "dgrad 2p", "os dgrad", "ratio"))
for s, Cin, Cout in cases:
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    perm, wm, order, t_sort = sorted_rows(m)
    blocks = sum(bin(int(v) & 0xFFFFFFFF).count("1") for v in wm.tolist())
    fill = m.P / (32.0 * blocks)
    x = torch.randn(n, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    g = torch.randn(n, Cout, device="cuda"); Wt = W.transpose(1, 2).contiguous()
    bias = torch.randn(Cout, device="cuda")
    T = torch.empty(m.P, Cout, device="cuda"); out = torch.empty(n, Cout, device="cuda"); out_os = torch.empty_like(out)
    rp, rl = m.rows("out")
    t_g = timeit(lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None))
    t_r = timeit(lambda: call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, Cout, ptr(bias), None, ptr(out)))
    t_o = timeit(lambda: call("lidog_sconv_os", ptr(x), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(W), 0, ptr(bias), None, Cin, Cout, ptr(out_os)))
    assert not CHECK or torch.equal(out, out_os), f"forward differs: {(out - out_os).abs().max().item()}"
    # data gradient with a residual addend
    T2 = torch.empty(m.P, Cin, device="cuda"); gx = torch.empty(n, Cin, device="cuda"); gx_os = torch.empty_like(gx)
    ad = torch.randn(n, Cin, device="cuda")
    rpi, rli = m.rows("in")
    t_dg = timeit(lambda: ME._gemm(g, m.pair_out, Wt, None, m, Cout, Cin, T2, None))
    t_dr = timeit(lambda: call("lidog_sconv_reduce_rows", ptr(T2), ptr(rpi), ptr(rli), n, Cin, None, ptr(ad), ptr(gx)))
    t_do = timeit(lambda: call("lidog_sconv_os", ptr(g), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(Wt), 1, None, ptr(ad), Cout, Cin, ptr(gx_os)))
    assert not CHECK or torch.equal(gx, gx_os), f"data gradient differs: {(gx - gx_os).abs().max().item()}"
    fl = 2.0 * m.P * Cin * Cout / 1e9
    print("s%-2d %3d->%3d      %9d %6.3f | %8.3f %8.3f %8.3f | %8.3f %7.1f %6.2f | %8.3f %8.3f %6.2f   (sort %.3f ms)" % (
        s, Cin, Cout, m.P, fill, t_g, t_r, t_g + t_r, t_o, fl / t_o, (t_g + t_r) / t_o, t_dg + t_dr, t_do, (t_dg + t_dr) / t_do, t_sort))

# ---- statistics forms against the two-pass forms (sums in another order: equal to rounding, results identical)
print("statistics epilogues:")
for s, Cin, Cout in [(1, 96, 96), (2, 32, 32), (1, 128, 96)]:
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    perm, wm, order, _ = sorted_rows(m)
    x = torch.randn(n, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda")
    rp, rl = m.rows("out")
    def fwd_two():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
        call("lidog_sconv_reduce_rows_stats", ptr(T), ptr(rp), ptr(rl), n, Cout, None, ptr(o1), ptr(su1), ptr(ws1), float(n), 1e-5, 0.1,
             ptr(me1), ptr(is1), None, None)
    def fwd_os():
        call("lidog_sconv_os_stats", ptr(x), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(W), None, Cin, Cout, ptr(o2),
             ptr(su2), ptr(ws2), float(n), 1e-5, 0.1, ptr(me2), ptr(is2), None, None)
    o1 = torch.empty(n, Cout, device="cuda"); o2 = torch.empty_like(o1)
    su1 = torch.empty(2 * Cout + 1, dtype=torch.float64, device="cuda"); su2 = torch.empty_like(su1)
    ws1 = torch.empty(L.lidog_sconv_reduce_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
    ws2 = torch.empty(L.lidog_sconv_os_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
    me1 = torch.empty(Cout, device="cuda"); is1 = torch.empty(Cout, device="cuda"); me2 = torch.empty(Cout, device="cuda"); is2 = torch.empty(Cout, device="cuda")
    t1, t2 = timeit(fwd_two), timeit(fwd_os)
    ref = torch.cat([o1.double().sum(0), (o1.double() ** 2).sum(0)])
    assert torch.equal(o1, o2)
    e1 = ((su1[:-1] - ref).abs() / ref.abs().clamp_min(1e-3)).max().item(); e2 = ((su2[:-1] - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    assert e2 < 1e-9 and su2[-1].item() == n and torch.allclose(me1, me2, rtol=1e-6, atol=1e-7) and torch.allclose(is1, is2, rtol=1e-6), (e1, e2)
    print("s%d %3d->%3d  fwd+stats two-pass %.3f os %.3f (%.2fx)" % (s, Cin, Cout, t1, t2, t1 / t2))
print("ok")
