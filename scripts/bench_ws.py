"""Gathered GEMM, default kernel vs the wave-specialised one (LIDOG_GEMM_WS=1, read once per process): per-layer time and a
checksum of the product rows (the two runs must print the same checksums)."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth

b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
print("LIDOG_GEMM_WS =", os.environ.get("LIDOG_GEMM_WS", "0"))
tot = 0.0
for s, Cin, Cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 256, 256),
                     (8, 128, 128), (8, 384, 256), (16, 256, 256), (4, 192, 128)):
    m = cm.kernel_map(s, s, 3)
    g = torch.Generator(device="cuda").manual_seed(s * 1000 + Cin)
    x = torch.randn(m.n_in, Cin, device="cuda", generator=g)
    W = torch.randn(m.K, Cin, Cout, device="cuda", generator=g) * 0.1
    T = torch.zeros(m.P, Cout, device="cuda")
    for _ in range(3):
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    h = hashlib.sha1(T.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"s{s:<2d} {Cin:3d}->{Cout:3d}  P {m.P:8d}  {ms:.3f} ms  {2e-9 * m.P * Cin * Cout / ms:6.1f} TF/s  T sha1 {h}")
print(f"sum {tot:.3f} ms")
