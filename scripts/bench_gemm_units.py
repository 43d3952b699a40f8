"""Gathered GEMM alone, one unit per workgroup against several (csrc/sconv_mfma.hip:k_sconv_gemm_mfma_ms), on the bench
workload's real kernel maps:  BS=4 python scripts/bench_gemm_units.py   (the two forms must agree bit for bit)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib

bs = int(os.environ.get("BS", 4))
b = synth.make_batch(range(bs), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
L = _lib.load()


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


cases = [(1, 3, 96, 96), (2, 3, 96, 96), (4, 3, 128, 128), (4, 3, 64, 64), (4, 3, 192, 128), (8, 3, 128, 128), (8, 3, 256, 256),
         (8, 3, 384, 256), (16, 3, 256, 256), (16, 3, 128, 256), (1, 1, 128, 96), (4, 1, 192, 128), (8, 1, 384, 256), (16, 1, 128, 256)]
print(f"bs {bs}   {'layer':18s} {'units':>7s} {'one ms':>8s} {'TF/s':>6s} | {'multi ms':>8s} {'TF/s':>6s} {'gain':>6s}")
tot = [0.0, 0.0]
for s, k, Cin, Cout in cases:
    if k == 1:
        n = cm.kernel_map(s, s, 3).n_out
        m = ME._IdentityMap(n, "cuda")
        gather = None
    else:
        m = cm.kernel_map(s, s, k)
        gather = m.pair_in
    x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = [torch.empty(m.P, Cout, device="cuda") for _ in range(2)]
    t = []
    for i, multi in enumerate((0, 1)):
        L.lidog_sconv_gemm_units(multi, 0)
        t.append(timeit(lambda: ME._gemm(x, gather, W, None, m, Cin, Cout, T[i], None)))
    L.lidog_sconv_gemm_units(1, 0)
    same = torch.equal(T[0], T[1])
    nt = 128 if Cout % 128 == 0 else 96 if Cout % 96 == 0 else 64 if Cout % 64 == 0 else 32
    fl = 2.0 * m.P * Cin * Cout / 1e9
    print(f"s{s:<2d} k{k} {Cin:3d}->{Cout:3d}        {m.n_tiles * (Cout // nt):7d} {t[0]:8.3f} {fl / t[0]:6.1f} | {t[1]:8.3f} {fl / t[1]:6.1f} {100 * (1 - t[1] / t[0]):5.1f}%"
          + ("" if same else "   BITS DIFFER"))
    tot[0] += t[0]; tot[1] += t[1]
print(f"sum ms: one unit {tot[0]:.3f}   several {tot[1]:.3f}")
