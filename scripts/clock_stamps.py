"""In-kernel shader clock of the matrix kernels (diagnostic build, csrc/clock_stamp.h):
    LIDOG_SO=<variant built with -DLIDOG_CLOCK_STAMP> python scripts/clock_stamps.py gemm|wgrad|os|conv2d
Each case runs ~2 s of back-to-back launches on random data (the chip settles on the clock it holds under that load),
then reads the per-workgroup stamps of the last launches: clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz,
median over workgroups (MI355X_MICROARCH.md, DVFS give-back item 6).  Also prints the launch's wall time and TF/s."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr

what = sys.argv[1]
L = _lib.load()
raw = ctypes.CDLL(_lib.SO_PATH)
SLOTS = 4096


def stamps():
    buf = (ctypes.c_ulonglong * (4 * SLOTS))()
    assert raw.lidog_debug_clock_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(SLOTS, 4).astype(np.float64)
    a = a[a[:, 3] > 0]
    a = a[a[:, 1] > 50]          # workgroups shorter than 0.5 us say nothing
    return a[:, 0] / a[:, 1] * 0.1 if len(a) else np.array([])      # GHz


def run(name, fn, flops, seconds=2.0):
    fn(); torch.cuda.synchronize()
    stamps()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    g = stamps()
    if len(g):
        q = np.percentile(g, [10, 50, 90])
        tf = flops / ms / 1e9
        roof = 157.3 * q[1] / 2.4
        print(f"{name:34s} {ms:8.3f} ms {tf:7.1f} TF/s | clock GHz p10 {q[0]:.3f} median {q[1]:.3f} p90 {q[2]:.3f} ({len(g)} workgroups) | "
              f"of 157.3: {tf / 157.3:.3f}  of the roof at that clock ({roof:.1f}): {tf / roof:.3f}", flush=True)
    else:
        print(f"{name:34s} {ms:8.3f} ms  no stamps (kernel not in this variant)", flush=True)


bs = int(os.environ.get("BS", 4))
if what in ("gemm", "wgrad", "os"):
    b = synth.make_batch(range(bs), "kitti120k", "cuda")
    st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
    cm = st.coordinate_manager
    prev = 1
    for s in (2, 4, 8, 16):
        cm.stride(prev, s); prev = s
if what == "gemm":
    for s, Cin, Cout in ((1, 96, 96), (2, 96, 96), (4, 128, 128), (8, 256, 256), (8, 128, 128), (16, 256, 256), (4, 64, 64)):
        m = cm.kernel_map(s, s, 3)
        x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
        T = torch.empty(m.P, Cout, device="cuda")
        for multi in (0, 1):
            L.lidog_sconv_gemm_units(multi, 0)
            run(f"gemm s{s} {Cin}->{Cout} units={'n' if multi else '1'}", lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None),
                2.0 * m.P * Cin * Cout)
elif what == "wgrad":
    for s, Cin, Cout in ((1, 96, 96), (2, 96, 96), (4, 128, 128), (8, 256, 256), (16, 256, 256)):
        m = cm.kernel_map(s, s, 3)
        x = torch.randn(m.n_in, Cin, device="cuda"); g = torch.randn(m.n_out, Cout, device="cuda")
        gW = torch.empty(m.K, Cin, Cout, device="cuda")
        items, ns, item_off = ME._wgrad_items(m, Cin, Cout)
        part = torch.empty(max(L.lidog_sconv_wgrad_slabs(Cin, Cout, ns), 1), Cin, Cout, device="cuda")
        run(f"wgrad s{s} {Cin}x{Cout}", lambda: call("lidog_sconv_wgrad", ptr(x), ptr(m.pair_in), ptr(g), ptr(m.pair_out), ptr(items), ns,
                                                      ptr(item_off), m.K, Cin, Cout, ptr(part), ptr(gW)), 2.0 * m.P * Cin * Cout)
elif what == "os":
    for s, Cin, Cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32)):
        m = cm.kernel_map(s, s, 3)
        n = m.n_out
        pad = (n + 127) // 128 * 128
        perm = torch.empty(pad, dtype=torch.int32, device="cuda"); wm = torch.empty(pad // 32, dtype=torch.int32, device="cuda")
        order = torch.empty(pad // 128, dtype=torch.int32, device="cuda")
        ws = torch.empty(L.lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device="cuda")
        call("lidog_kernel_map_sorted", ptr(m.nbr), n, m.K, ptr(m.k_off), ptr(perm), ptr(wm), ptr(order), ptr(ws), ws.numel())
        x = torch.randn(n, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
        out = torch.empty(n, Cout, device="cuda")
        run(f"os fwd s{s} {Cin}->{Cout}", lambda: call("lidog_sconv_os", ptr(x), ptr(m.nbr), n, m.K, ptr(perm), ptr(wm), ptr(order), ptr(W), 0,
                                                        None, None, Cin, Cout, ptr(out)), 2.0 * m.P * Cin * Cout)
elif what == "conv2d":
    B = bs
    for Cin, H, Cout in ((256, 333, 256),):
        Ho = (H + 2 - 3) // 2 + 1
        x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        y = torch.empty(B, Cout, Ho, Ho, device="cuda"); gy = torch.randn_like(y); gx = torch.empty_like(x)
        wsd = torch.empty(9 * Cin * Cout, device="cuda")
        fl = 2.0 * B * Ho * Ho * Cout * Cin * 9
        run(f"conv2d fwd {Cin}->{Cout} {H}", lambda: call("lidog_conv2d_fwd", ptr(x), ptr(w), None, B, Cin, H, H, Cout, 3, 2, 1, ptr(y)), fl)
        run(f"conv2d dgrad {Cin}->{Cout} {H}", lambda: call("lidog_conv2d_dgrad", ptr(gy), ptr(w), B, Cin, H, H, Cout, 3, 2, 1, ptr(gx), ptr(wsd)), fl)
