"""One gathered-GEMM layer of the bench workload, a few launches (for rocprofv3 --pmc):  S CIN COUT from the environment"""
import os, sys
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
s, Cin, Cout = int(os.environ.get("S", 1)), int(os.environ.get("CIN", 96)), int(os.environ.get("COUT", 96))
m = cm.kernel_map(s, s, 3)
x = torch.randn(m.n_in, Cin, device="cuda"); W = torch.randn(m.K, Cin, Cout, device="cuda") * 0.1
T = torch.empty(m.P, Cout, device="cuda")
for _ in range(5):
    ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
torch.cuda.synchronize()
