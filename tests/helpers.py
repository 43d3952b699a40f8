"""Shared test helpers: deterministic weights and small synthetic inputs.

`seeded_state_dict` regenerates identical parameter values on any box with the
same torch build, so golden fixtures only carry inputs and expected outputs
(a MinkUNet34BEV state_dict is 155 MB)."""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def seeded_state_dict(model, seed=0):
    sd = model.state_dict()
    out = {}
    for idx, (name, t) in enumerate(sd.items()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if name.endswith("num_batches_tracked"):
            v = torch.zeros_like(t)
        elif name.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif name.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif t.dim() == 1 and name.endswith("weight"):
            v = torch.rand(t.shape, generator=g) * 0.6 + 0.4   # BN gains < 1 keep the residual stack O(1)
        elif name.endswith("bias"):
            v = torch.randn(t.shape, generator=g) * 0.1
        else:
            if t.dim() == 3:      # ME kernel [K,Cin,Cout]
                fan = t.shape[1] * min(t.shape[0], 8)
            elif t.dim() == 2:    # ME 1x1 kernel [Cin,Cout]
                fan = t.shape[0]
            else:                 # Conv2d [Cout,Cin,kh,kw]
                fan = t.shape[1] * t.shape[2] * t.shape[3]
            v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan)
            if name == "final.kernel":
                v = v * 0.25                                   # logits O(1): 1e-4 absolute is a real bar
        out[name] = v.to(t.dtype)
    return out


def small_scene(seed, n_points=3000, extent=4.8, voxel=0.05, oob=40):
    """A ground sheet + a few vertical walls inside +-extent metres, plus `oob`
    points just outside +-5 m (exercise sparse2super's bounds filter at B=5).
    Returns unique int32 voxel coords [n,3] in first-occurrence order."""
    rng = np.random.default_rng(seed)
    n_g = n_points // 2
    ground = np.stack([rng.uniform(-extent, extent, n_g), rng.uniform(-extent, extent, n_g),
                       -1.0 + 0.03 * rng.standard_normal(n_g)], axis=1)
    walls = []
    for _ in range(6):
        x0, y0 = rng.uniform(-extent, extent, 2)
        ang = rng.uniform(0, math.pi)
        t = rng.uniform(0, 2.0, (n_points - n_g) // 6)
        z = rng.uniform(-1.0, 1.5, t.shape[0])
        walls.append(np.stack([np.clip(x0 + t * math.cos(ang), -extent, extent),
                               np.clip(y0 + t * math.sin(ang), -extent, extent), z], axis=1))
    out_pts = np.stack([rng.choice([-1, 1], oob) * rng.uniform(5.0, 5.4, oob), rng.uniform(-5.4, 5.4, oob),
                        rng.uniform(-1, 1, oob)], axis=1)
    pts = np.concatenate([ground] + walls + [out_pts], axis=0).astype(np.float32)
    pts = pts[rng.permutation(pts.shape[0])]
    vox = np.floor(pts / voxel).astype(np.int32)
    _, first = np.unique(vox, axis=0, return_index=True)
    return vox[np.sort(first)]


def small_batch(seeds=(0, 1), **kw):
    """Collated batch like CollateFNSingleSourceBEVMultiLevel: coords int32 [N,4] with batch column."""
    cs = []
    for b, s in enumerate(seeds):
        v = small_scene(s, **kw)
        cs.append(np.concatenate([np.full((v.shape[0], 1), b, np.int32), v], axis=1))
    return torch.from_numpy(np.concatenate(cs, axis=0))


def sha_triples(k_off, pin, pout):
    """Order-independent fingerprint of a kernel map: sha1 of the sorted (k,in,out) triples."""
    import hashlib
    k_off = np.asarray(k_off, dtype=np.int64)
    ks = np.repeat(np.arange(len(k_off) - 1, dtype=np.int64), np.diff(k_off))
    tri = np.stack([ks, np.asarray(pin, np.int64), np.asarray(pout, np.int64)], axis=1)
    tri = tri[np.lexsort((tri[:, 2], tri[:, 1], tri[:, 0]))]
    return hashlib.sha1(np.ascontiguousarray(tri).tobytes()).hexdigest()
