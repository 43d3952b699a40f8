"""Gathered GEMM: default kernel vs the LDS-DMA pipeline experiment (csrc/sconv_dma.hip, 256-row tiles): per-layer time and
bit-identity of the product rows."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr

b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s


def tiles256(m):
    k_off = np.ascontiguousarray(m.k_off_host, dtype=np.int64)
    K = len(k_off) - 1
    cap = int(k_off[-1]) // 256 + K + 1
    out = np.empty(3 * cap, dtype=np.int32)
    n = _lib.load().lidog_tiles_host(k_off.ctypes.data, K, -1, 256, out.ctypes.data, cap)
    return torch.from_numpy(out[:3 * n].reshape(3, n).copy()).cuda(), int(n)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


layers = ((1, 96, 96), (1, 128, 96), (2, 96, 96), (4, 128, 128), (4, 64, 64), (8, 256, 256), (8, 128, 128),
          (8, 384, 256), (16, 256, 256), (4, 192, 128))
only = os.environ.get("ONLY")
tot0 = tot1 = 0.0
for li, (s, Cin, Cout) in enumerate(layers):
    if only is not None and int(only) != li:
        continue
    m = cm.kernel_map(s, s, 3)
    g = torch.Generator(device="cuda").manual_seed(s * 1000 + Cin)
    x = torch.randn(m.n_in, Cin, device="cuda", generator=g)
    W = torch.randn(m.K, Cin, Cout, device="cuda", generator=g) * 0.1
    T0 = torch.zeros(m.P, Cout, device="cuda"); T1 = torch.zeros(m.P, Cout, device="cuda")
    desc, n_t = tiles256(m)
    f0 = lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T0, None)
    f1 = lambda: call("lidog_sconv_gemm_dma", ptr(x), ptr(m.pair_in), ptr(W), ptr(desc[0]), ptr(desc[1]), ptr(desc[2]), n_t,
                      Cin, Cout, ptr(T1))
    f0(); f1(); torch.cuda.synchronize()
    same = torch.equal(T0, T1)
    bad = int((T0 != T1).any(dim=1).sum()) if not same else 0
    t0, t1 = timeit(f0), timeit(f1)
    tot0 += t0; tot1 += t1
    fl = 2e-9 * m.P * Cin * Cout
    print(f"s{s:<2d} {Cin:3d}->{Cout:3d}  default {t0:.3f} ms {fl / t0:6.1f} TF/s | dma {t1:.3f} ms {fl / t1:6.1f} TF/s | "
          f"bit-identical {same}" + ("" if same else f" ({bad} of {m.P} rows differ)"), flush=True)
print(f"sum: default {tot0:.3f} ms, dma {tot1:.3f} ms")
