"""VERDICT r5 item 2, measured: the output-stationary 3^3 convolution with the offsets of a tile cut into 3 ascending
groups of about equal pair count (3 workgroups per 128-row tile, partial slabs added in order by a second kernel) against
the exact output-stationary kernel and the two-pass path (gathered GEMM + per-row reduction), forward and data gradient, on
the bench scans' maps.  Needs the experimental build:
    so=$(bash scripts/build_variant.sh os_split sconv_os.hip -DLIDOG_EXP_OS_SPLIT | tail -1)
    BS=2 LIDOG_SO=$PWD/$so python scripts/bench_os_split.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
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
raw = ctypes.CDLL(_lib.SO_PATH)
P = ctypes.c_void_p
raw.lidog_exp_sconv_os_split.argtypes = [P, P, ctypes.c_int64, ctypes.c_int32, P, P, P, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         P, P, ctypes.c_int32, P, P]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sorted_rows(m):
    n = m.n_out
    pad = (n + 127) // 128 * 128
    perm = torch.empty(pad, dtype=torch.int32, device="cuda"); wm = torch.empty(pad // 32, dtype=torch.int32, device="cuda")
    order = torch.empty(pad // 128, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device="cuda")
    call("lidog_kernel_map_sorted", ptr(m.nbr), n, m.K, ptr(m.k_off), ptr(perm), ptr(wm), ptr(order), ptr(ws), ws.numel())
    return perm, wm, order


def group_masks(k_off, groups=3):
    cnt = np.diff(np.asarray(k_off, dtype=np.int64))
    cum = np.cumsum(cnt); total = cum[-1]
    masks, k0 = [], 0
    for g in range(groups):
        k1 = int(np.searchsorted(cum, total * (g + 1) / groups, side="left")) + 1 if g + 1 < groups else len(cnt)
        k1 = max(k1, k0 + 1)
        masks.append(sum(1 << k for k in range(k0, min(k1, len(cnt)))))
        k0 = min(k1, len(cnt))
    return np.array(masks, dtype=np.uint32)


cases = [(1, 96, 96), (2, 96, 96), (4, 128, 128), (4, 64, 64), (8, 256, 256), (8, 128, 128), (16, 256, 256)]
stream = torch.cuda.current_stream().cuda_stream
print(f"bs {bs}  {'layer':14s} {'tiles':>6s} | fwd: {'two-pass':>8s} {'os exact':>8s} {'os split':>8s} | dgrad: {'two-pass':>8s} {'os exact':>8s} {'os split':>8s} | max rel diff split vs exact")
for s, Cin, Cout in cases:
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    perm, wm, order = sorted_rows(m)
    masks = torch.from_numpy(group_masks(m.k_off_host).view(np.int32)).cuda()
    masks_host = (ctypes.c_uint32 * 3)(*[int(v) & 0xFFFFFFFF for v in group_masks(m.k_off_host)])
    x = torch.randn(n, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    g = torch.randn(n, Cout, device="cuda"); Wt = W.transpose(1, 2).contiguous()
    T = torch.empty(m.P, max(Cin, Cout), device="cuda")
    out = torch.empty(n, Cout, device="cuda"); out_os = torch.empty_like(out); out_sp = torch.empty_like(out)
    gx = torch.empty(n, Cin, device="cuda"); gx_os = torch.empty_like(gx); gx_sp = torch.empty_like(gx)
    part = torch.empty(3, n, max(Cin, Cout), device="cuda")
    rp, rl = m.rows("out"); rpi, rli = m.rows("in")

    def two_f():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T[:, :Cout].contiguous() if False else T.view(-1)[:m.P * Cout].view(m.P, Cout), None)
        call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, Cout, None, None, ptr(out))

    def two_d():
        ME._gemm(g, m.pair_out, Wt, None, m, Cout, Cin, T.view(-1)[:m.P * Cin].view(m.P, Cin), None)
        call("lidog_sconv_reduce_rows", ptr(T), ptr(rpi), ptr(rli), n, Cin, None, None, ptr(gx))

    def os_f():
        call("lidog_sconv_os", ptr(x), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(W), 0, None, None, Cin, Cout, ptr(out_os))

    def os_d():
        call("lidog_sconv_os", ptr(g), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(Wt), 1, None, None, Cout, Cin, ptr(gx_os))

    def sp_f():
        rc = raw.lidog_exp_sconv_os_split(ptr(x), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(W), 0, Cin, Cout, ptr(part),
                                          ctypes.cast(masks_host, P), 3, ptr(out_sp), stream)
        assert rc == 0

    def sp_d():
        rc = raw.lidog_exp_sconv_os_split(ptr(g), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(Wt), 1, Cout, Cin, ptr(part),
                                          ctypes.cast(masks_host, P), 3, ptr(gx_sp), stream)
        assert rc == 0

    t = [timeit(f) for f in (two_f, os_f, sp_f, two_d, os_d, sp_d)]
    assert torch.equal(out, out_os) and torch.equal(gx, gx_os)
    rel = max(((out_sp - out_os).abs().max() / out_os.abs().max()).item(), ((gx_sp - gx_os).abs().max() / gx_os.abs().max()).item())
    print(f"s{s:<2d} {Cin:3d}->{Cout:3d}     {(n + 127) // 128:6d} |      {t[0]:8.3f} {t[1]:8.3f} {t[2]:8.3f} |        {t[3]:8.3f} {t[4]:8.3f} {t[5]:8.3f} | {rel:.2e}", flush=True)
