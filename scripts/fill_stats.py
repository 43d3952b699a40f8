"""How much MFMA work would an output-stationary (T-free) sparse convolution waste?  For every stride level,
for the 3x3x3 map: pairs P vs 32-row-group x offset occupancy (rows in natural order and Morton-sorted)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import oracle.me_cpu as OME
from lidog_amd import synth
config = sys.argv[1] if len(sys.argv) > 1 else "kitti120k"
vox, _ = synth.scan_voxels(0, config)
C = torch.from_numpy(np.concatenate([np.zeros((vox.shape[0], 1), np.int32), vox], axis=1))
st = OME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1)))
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
def morton(c, s):
    c = (c[:, 1:].astype(np.int64) // s) + 4096
    key = np.zeros(c.shape[0], np.int64)
    for b in range(13):
        for d in range(3):
            key |= ((c[:, d] >> b) & 1) << (3 * b + d)
    return np.argsort(key, kind="stable")
for s in (1, 2, 4, 8, 16):
    k_off, pin, pout, nbr = cm.kernel_map(s, s, 3)
    nbr = nbr.numpy(); N = nbr.shape[0]; P = int((nbr >= 0).sum())
    res = []
    for G in (16, 32):
        for name, order in (("natural", np.arange(N)), ("morton", morton(cm.maps[s].numpy(), s))):
            occ = (nbr[order] >= 0)
            pad = (-N) % G
            occ = np.concatenate([occ, np.zeros((pad, 27), bool)])
            g = occ.reshape(-1, G, 27).any(axis=1).sum() * G
            res.append(f"{name}{G}: {g / P:.2f}x")
    print(f"stride {s:2d}: N={N:6d} P={P:8d} P/N={P/N:5.2f}  dense={27*N/P:.2f}x  " + "  ".join(res))
