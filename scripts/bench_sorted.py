"""time of lidog_kernel_map_sorted (rows sorted by neighbour mask: hand-written radix sort + tile order) on the stride-1 and
stride-2 3^3 maps of the bench batch"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr
b = synth.make_batch(range(4), "kitti120k", "cuda")
cm = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]).coordinate_manager
cm.stride(1, 2)
for s in (1, 2):
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    pad = (n + 127) // 128 * 128
    perm = torch.empty(pad, dtype=torch.int32, device="cuda"); wm = torch.empty(pad // 32, dtype=torch.int32, device="cuda")
    order = torch.empty(pad // 128, dtype=torch.int32, device="cuda")
    ws = torch.empty(_lib.load().lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device="cuda")
    f = lambda: call("lidog_kernel_map_sorted", ptr(m.nbr), n, m.K, ptr(m.k_off), ptr(perm), ptr(wm), ptr(order), ptr(ws), ws.numel())
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"stride {s}: {n} rows, {pad // 128} tiles: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per call")
