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
cases = [(1, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 256, 256), (8, 384, 256), (16, 256, 256), (2, 96, 96)]
for target in (512, 1024, 2048, 4096, 8192):
    ME._WGRAD_TARGET_BLOCKS = target
    tot = 0; row = []
    for s, Cin, Cout in cases:
        m = cm.kernel_map(s, s, 3); m.__dict__.pop("_wgrad_items", None)
        x = torch.randn(m.n_in, Cin, device="cuda"); g = torch.randn(m.n_out, Cout, device="cuda"); gW = torch.empty(m.K, Cin, Cout, device="cuda")
        items, n, off = ME._wgrad_items(m, Cin, Cout)
        part = torch.empty(max(L.lidog_sconv_wgrad_slabs(Cin, Cout, n), 1), Cin, Cout, device="cuda")
        t = timeit(lambda: call("lidog_sconv_wgrad", ptr(x), ptr(m.pair_in), ptr(g), ptr(m.pair_out), ptr(items), n, ptr(off), m.K, Cin, Cout, ptr(part), ptr(gW)))
        tot += t; row.append("%d:%.3f" % (n, t))
    print("target %5d  sum %.3f ms  " % (target, tot), " ".join(row))
