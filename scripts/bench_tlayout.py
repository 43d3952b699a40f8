"""Experiment: product rows T written in per-output-row (CSR) order instead of offset-major order -- the reduction then
reads every output row's products as one contiguous run.  Uses the existing kernels: lidog_sconv_gemm with a scatter
index (CSR slot of every pair) and lidog_sconv_reduce_rows with row_list = 0..P-1."""
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


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("layer            gemm(k-major T)  gemm(CSR T)   reduce(list)  reduce(contiguous)   pair: before -> after")
for s, Cin, Cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (8, 256, 256)):
    m = cm.kernel_map(s, s, 3)
    x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda"); out = torch.empty(m.n_out, Cout, device="cuda")
    row_ptr, row_list = m.rows("out")
    inv = torch.empty(m.P, dtype=torch.int32, device="cuda")
    inv[row_list[:m.P].long()] = torch.arange(m.P, dtype=torch.int32, device="cuda")
    ident = torch.arange(m.P, dtype=torch.int32, device="cuda")
    g0 = timeit(lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None))
    r0 = timeit(lambda: call("lidog_sconv_reduce_rows", ptr(T), ptr(row_ptr), ptr(row_list), m.n_out, Cout, None, None, ptr(out)))
    ref = out.clone()
    g1 = timeit(lambda: ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, inv))
    r1 = timeit(lambda: call("lidog_sconv_reduce_rows", ptr(T), ptr(row_ptr), ptr(ident), m.n_out, Cout, None, None, ptr(out)))
    same = torch.equal(out, ref)
    print(f"s{s} {Cin:3d}->{Cout:3d}      {g0:.3f}          {g1:.3f}         {r0:.3f}         {r1:.3f}            {g0 + r0:.3f} -> {g1 + r1:.3f}  bit-identical {same}")
