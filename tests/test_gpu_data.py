"""GPU parity of the device-side data path: voxelisation / collation against the oracle's
ME.utils.sparse_quantize, BEV label rasteriser against images produced by the reference's own
PC2ImgConverter.getBEVImageNew (tests/golden/g7_bev_labels.npz)."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("q", [0.05, 0.1, (0.05, 0.1, 0.2)])
def test_sparse_quantize_matches_oracle(q):
    import oracle.me_cpu as OME
    from lidog_amd.data import sparse_quantize
    from lidog_amd import synth
    pts, rng = synth.scan_points(3, **synth.CONFIGS["source8k"])
    pts = np.concatenate([pts, pts[:500] + np.float32(0.004)])       # several points per voxel
    labels = rng.integers(-1, 7, pts.shape[0])
    feats = rng.standard_normal((pts.shape[0], 2)).astype(np.float32)
    ref = OME.utils.sparse_quantize(pts, feats, labels=labels, ignore_label=-1, quantization_size=q,
                                    return_index=True, return_inverse=True)
    got = sparse_quantize(torch.from_numpy(pts).cuda(), torch.from_numpy(feats).cuda(),
                          labels=torch.from_numpy(labels).cuda(), ignore_label=-1, quantization_size=q,
                          return_index=True, return_inverse=True)
    assert len(got) == len(ref) == 5
    for r, g in zip(ref, got):
        assert np.array_equal(np.asarray(r), g.cpu().numpy())
    assert int((got[2] == -1).sum()) > 0  # disagreeing points really produced ignore labels
    only = sparse_quantize(torch.from_numpy(pts).cuda(), quantization_size=q)
    assert torch.equal(only, got[0])


def test_me_utils_sparse_quantize_has_minkowski_engine_call_shapes():
    """the zero-edit alias route: ME.utils.sparse_quantize as the reference calls it -- numpy in / numpy out with 5
    results (semantickitti_bev.py:232-238), 4 (mix3D.py:67-72), 3 with a vector quantization_size
    (minkunet_bev.py:279-284), CPU tensors in / CPU tensors out -- equal to the oracle's"""
    import oracle.me_cpu as OME
    import lidog_amd.me as ME
    from lidog_amd import synth
    pts, rng = synth.scan_points(5, **synth.CONFIGS["source8k"])
    pts = np.concatenate([pts, pts[:300] + np.float32(0.004)])
    labels = rng.integers(-1, 7, pts.shape[0])
    feats = np.ones((pts.shape[0], 1), np.float32)
    calls = [dict(args=(pts, feats), kw=dict(labels=labels, ignore_label=-1, quantization_size=0.05, return_index=True,
                                             return_inverse=True), n=5),
             dict(args=(pts, feats), kw=dict(labels=labels, ignore_label=-1, quantization_size=0.05, return_index=True), n=4),
             dict(args=(pts,), kw=dict(quantization_size=[3.0, 2.0, 19.0], return_index=True, return_inverse=True), n=3)]
    for c in calls:
        ref = OME.utils.sparse_quantize(*c["args"], **c["kw"])
        got = ME.utils.sparse_quantize(*c["args"], **c["kw"])
        assert len(got) == len(ref) == c["n"]
        for r, g in zip(ref, got):
            assert isinstance(g, np.ndarray) and g.dtype == np.asarray(r).dtype and np.array_equal(np.asarray(r), g)
    t = ME.utils.sparse_quantize(torch.from_numpy(pts), quantization_size=0.1, return_index=True)
    r = OME.utils.sparse_quantize(torch.from_numpy(pts), quantization_size=0.1, return_index=True)
    assert all(torch.is_tensor(a) and a.device.type == "cpu" and torch.equal(a, b) for a, b in zip(t, r))
    maps = ME.utils.sparse_quantize(pts, quantization_size=0.05, return_maps_only=True, return_inverse=True)
    assert np.array_equal(maps[0], OME.utils.sparse_quantize(pts, quantization_size=0.05, return_maps_only=True))


def test_collate_matches_sparse_collation():
    import oracle.me_cpu as OME
    from lidog_amd.data import collate
    g = torch.Generator().manual_seed(0)
    scans = [(torch.randint(-50, 50, (n, 3), generator=g, dtype=torch.int32), torch.ones(n, 1),
              torch.randint(-1, 7, (n,), generator=g)) for n in (100, 57, 311)]
    ref = OME.utils.SparseCollation(dtype=torch.float32)([(c, f, l) for c, f, l in scans])
    got = collate([(c.cuda(), f.cuda(), l.cuda()) for c, f, l in scans])
    for r, x in zip(ref, got):
        assert r.dtype == x.dtype and torch.equal(r, x.cpu())


@pytest.mark.parametrize("bound,size", [(50.0, 167), (30.0, 100)])
def test_bev_label_rasteriser_matches_reference_golden(bound, size):
    from lidog_amd.data import bev_labels
    g7 = np.load(f"{GOLDEN}/g7_bev_labels.npz")
    tag = str(int(bound))
    vox, labels = g7[f"vox_{tag}"], g7[f"labels_{tag}"]
    # two scans in one batch: the golden scan and a reversed copy (different winners for colliding pixels)
    rev = vox[::-1].copy()
    coords = np.concatenate([np.concatenate([np.zeros((vox.shape[0], 1), np.int32), vox], 1),
                             np.concatenate([np.ones((vox.shape[0], 1), np.int32), rev], 1)])
    lab2 = np.concatenate([labels, labels[::-1]])
    img, idx = bev_labels(torch.from_numpy(coords).cuda(), torch.from_numpy(lab2).cuda(), bound=bound, img_size=size)
    assert img.dtype == torch.int64 and tuple(img.shape) == (2, size, size)
    assert np.array_equal(img[0].cpu().numpy(), g7[f"img_{tag}"])
    assert np.array_equal(idx[0].cpu().numpy(), g7[f"idx_{tag}"])
    # the reversed scan keeps the same occupied pixels; every winner is the last labelled row of its pixel
    assert np.array_equal((img[1] >= 0).cpu().numpy(), g7[f"img_{tag}"] >= 0)
    i1 = idx[1].cpu().numpy()
    occ = i1 >= 0
    assert np.array_equal(img[1].cpu().numpy()[occ], labels[::-1][i1[occ]])


def test_mix3d_merge_matches_reference_arithmetic():
    """utils/datasets/mix3D.py:44-87 with the oracle's sparse_quantize: float32 round trip of the coordinates,
    first-point labels"""
    import oracle.me_cpu as OME
    from lidog_amd.data import mix3d_merge
    from lidog_amd import synth
    a, la = synth.scan_voxels(4, "nusc35k")
    b, lb = synth.scan_voxels(5, "nusc35k")
    b[:200] = a[:200]                                   # guaranteed overlaps between the two scans
    fa, fb = np.ones((a.shape[0], 1), np.float32), 2 * np.ones((b.shape[0], 1), np.float32)
    coords = torch.cat([torch.from_numpy(a), torch.from_numpy(b)]) * 0.05      # int32 * float -> float32
    feats = torch.cat([torch.from_numpy(fa), torch.from_numpy(fb)])
    labels = torch.cat([torch.from_numpy(la), torch.from_numpy(lb)])
    q, _, _, idx = OME.utils.sparse_quantize(coords.numpy(), feats.numpy(), labels=labels.numpy(), ignore_label=-1,
                                             quantization_size=0.05, return_index=True)
    got = mix3d_merge({"coordinates": torch.from_numpy(a).cuda(), "features": torch.from_numpy(fa).cuda(),
                       "sem_labels": torch.from_numpy(la).cuda()},
                      {"coordinates": torch.from_numpy(b).cuda(), "features": torch.from_numpy(fb).cuda(),
                       "sem_labels": torch.from_numpy(lb).cuda()})
    assert np.array_equal(got["coordinates"].cpu().numpy(), q)
    assert np.array_equal(got["index"].cpu().numpy(), idx)
    assert torch.equal(got["sem_labels"].cpu(), labels[torch.from_numpy(idx)])
    assert torch.equal(got["features"].cpu(), feats[torch.from_numpy(idx)])
    assert q.shape[0] < a.shape[0] + b.shape[0]
