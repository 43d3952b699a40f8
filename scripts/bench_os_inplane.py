"""Feasibility of the in-plane / out-of-plane split: round 1's output-stationary kernel on the 9
in-plane offsets only (K = 9 sub-rule-book), the gathered GEMM on the 18 out-of-plane offsets only, against the two-pass
path on all 27."""
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
L = _lib.load()
BR = L.lidog_sconv_os_block_rows()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for s, Cin, Cout in [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64)]:
    m = cm.kernel_map(s, s, 3)
    n = m.n_out
    koh = np.asarray(m.k_off_host, dtype=np.int64)
    x = torch.randn(n, Cin, device="cuda"); W = torch.randn(27, Cin, Cout, device="cuda") * 0.1
    T = torch.empty(m.P, Cout, device="cuda"); out = torch.empty(n, Cout, device="cuda")
    rp, rl = m.rows("out")
    def two_pass():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
        call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, Cout, None, None, ptr(out))
    # in-plane sub-rule-book (offsets 9..17 are contiguous in the k-major pair arrays)
    p0, p1 = int(koh[9]), int(koh[18])
    pin, pout = m.pair_in[p0:p1].contiguous(), m.pair_out[p0:p1].contiguous()
    koff9 = torch.tensor((koh[9:19] - koh[9]).tolist(), dtype=torch.int64, device="cuda")
    nb = (n + BR - 1) // BR
    seg = torch.empty((9, nb + 1), dtype=torch.int32, device="cuda")
    call("lidog_sconv_os_segments", ptr(pout), ptr(koff9), 9, n, ptr(seg))
    W9 = W[9:18].contiguous()
    o9 = torch.empty(n, Cout, device="cuda")
    def os9():
        call("lidog_sconv_os", ptr(x), ptr(pin), ptr(pout), ptr(seg), 9, n, ptr(W9), 0, Cin, Cout, ptr(o9), None)
    # out-of-plane: tile list without offsets 9..17
    koh18 = koh.copy()
    cnt = np.diff(koh); cnt[9:18] = 0
    desc_all, _ = ME._tiles_host(m.k_off_host)
    keep = (desc_all[0] < 9) | (desc_all[0] > 17)
    desc = torch.from_numpy(np.ascontiguousarray(desc_all[:, keep])).cuda()
    tiles18 = (desc, int(keep.sum()), int(cnt.sum()))
    def gemm18():
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None, tiles18)
    t2, t9, t18 = timeit(two_pass), timeit(os9), timeit(gemm18)
    share = (p1 - p0) / m.P
    print("s%-2d %3d->%3d  in-plane share %.2f | two-pass(27) %.3f ms | OS(9 in-plane) %.3f + GEMM(18 sparse) %.3f = %.3f  (+ list walk of %.0f %% of T)"
          % (s, Cin, Cout, share, t2, t9, t18, t9 + t18, 100 * (1 - share)))
