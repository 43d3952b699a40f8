import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr
L = _lib.load()
b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
m = cm.kernel_map(1, 1, 3)
for Cin, Cout in ((96, 96), (128, 128), (32, 32)):
    x = torch.randn(m.n_in, Cin, device="cuda"); g = torch.randn(m.n_out, Cout, device="cuda"); gW = torch.empty(m.K, Cin, Cout, device="cuda")
    fl = 2.0 * m.P * Cin * Cout / 1e9
    for core in (1, 0):
        L.lidog_set_sparse_core(core)
        for ns in (1, 4, 10, 38, 100, 200):
            slabs = L.lidog_sconv_wgrad_slabs(Cin, Cout, ns)
            part = torch.empty(slabs, m.K, Cin, Cout, device="cuda")
            t = timeit(lambda: call("lidog_sconv_wgrad", ptr(x), ptr(m.pair_in), ptr(g), ptr(m.pair_out), ptr(m.k_off), m.K, Cin, Cout, ns, ptr(part), ptr(gW)))
            print(f"{Cin}->{Cout} core {core} ns {ns:4d} slabs {slabs:4d}: {t:.3f} ms  {fl / t:.1f} TF/s")
    # sequential (sorted) access instead of the rule book: identity pairs
    ident = torch.arange(m.P, dtype=torch.int32, device="cuda") % m.n_in
    L.lidog_set_sparse_core(1)
    part = torch.empty(L.lidog_sconv_wgrad_slabs(Cin, Cout, 38), m.K, Cin, Cout, device="cuda")
    t = timeit(lambda: call("lidog_sconv_wgrad", ptr(x), ptr(ident), ptr(g), ptr(ident), ptr(m.k_off), m.K, Cin, Cout, 38, ptr(part), ptr(gW)))
    print(f"{Cin}->{Cout} identity pairs ns 38: {t:.3f} ms {fl / t:.1f} TF/s")
print("k_off", m.k_off_host[:5], m.k_off_host[-3:])
