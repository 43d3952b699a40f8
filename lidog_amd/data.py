"""Device-side data path feeding the hot path.

  sparse_quantize  ME.utils.sparse_quantize as called at utils/datasets/semantickitti_bev.py:232-238
                   (floor(p / voxel) -> first point of every voxel, label vote, index + inverse maps)
  collate          ME.utils.SparseCollation (utils/collation/collation.py:309-310): batch index in column 0
  bev_labels       PC2ImgConverter.getBEVImageNew (utils/datasets/semantickitti_bev.py:433-464) on the voxel
                   coordinates (bev_points = quantized_coords * voxel_size, :244)

The reference runs these in DataLoader worker processes with numpy; at >50 scans/s per GPU that becomes the
bottleneck, and both are the same hash / winner-map kernels as the hot path."""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr


def sparse_quantize(points, features=None, labels=None, ignore_label=-100, quantization_size=0.05,
                    return_index=False, return_inverse=False):
    """points float32 [n,3] on the GPU -> same tuple layout as ME.utils.sparse_quantize:
    (coords int32 [m,3], [features[index]], [voxel_labels], [index], [inverse])."""
    _lib.require_gpu(points, "points")
    points = points.contiguous().float()
    n = points.shape[0]
    dev = points.device
    q = np.broadcast_to(np.asarray(quantization_size, dtype=np.float32), (3,))
    rows = torch.empty((n, 4), dtype=torch.int32, device=dev)
    call("lidog_voxel_floor", ptr(points), n, float(q[0]), float(q[1]), float(q[2]), 0, ptr(rows))
    cap = _lib.load().lidog_hash_capacity(n)
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    vals = torch.empty(cap, dtype=torch.int32, device=dev)
    first = torch.empty(n, dtype=torch.int32, device=dev)
    n_unique = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    call("lidog_coords_insert", ptr(rows), n, ptr(keys), ptr(vals), cap, ptr(first), ptr(n_unique), ptr(err))
    uniq = torch.empty(n, dtype=torch.int32, device=dev)
    inv = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(n + 2048, dtype=torch.int32, device=dev)
    call("lidog_coords_compact", ptr(first), n, ptr(keys), ptr(vals), cap, ptr(rows), ptr(uniq), ptr(inv), ptr(ws))
    m, bad = int(n_unique.item()), int(err.item())
    if bad:
        raise ValueError("voxel coordinates out of the supported range |c| <= 65535")
    uniq = uniq[:m]
    index = uniq.long()
    out = [rows[index][:, 1:].contiguous()]
    if features is not None:
        out.append(features[index])
    if labels is not None:
        lab = labels.to(torch.int32).contiguous()
        vlab = torch.empty(m, dtype=torch.int32, device=dev)
        call("lidog_label_vote", ptr(lab), ptr(uniq.contiguous()), ptr(inv), n, m, int(ignore_label), ptr(vlab))
        out.append(vlab.to(labels.dtype))
    if return_index:
        out.append(index)
    if return_inverse:
        out.append(inv.long())
    return out[0] if len(out) == 1 else tuple(out)


def collate(scans):
    """list of (coords int [n_i,3], feats [n_i,C], labels [n_i]) on the GPU -> (coords [N,4] float32 with the batch
    index in column 0, feats, labels), i.e. ME.utils.SparseCollation(dtype=torch.float32)."""
    coords = torch.cat([torch.cat([torch.full((c.shape[0], 1), b, dtype=c.dtype, device=c.device), c], dim=1)
                        for b, (c, _, _) in enumerate(scans)], dim=0)
    return coords.float(), torch.cat([f for _, f, _ in scans], dim=0), torch.cat([l for _, _, l in scans], dim=0)


_LABEL_LUTS = {}


def label_luts(bound, img_size, voxel, z_range=(-10.0, 8.0)):
    """pixel of an integer voxel coordinate under getBEVImageNew's float32 arithmetic:
    x = float32(c * voxel); lo < x < hi; px = floor((x - lo) / grid); py = floor(S - (y - lo) / grid) - 1."""
    key = (float(bound), int(img_size), float(voxel), tuple(z_range))
    if key not in _LABEL_LUTS:
        grid = (bound - (-bound)) / img_size                   # python float, as in semantickitti_bev.py:144-145
        S = int((bound - (-bound)) / grid)                     # maxImgWidth / maxImgHeight (:340-341)
        lo_c = -int(1.3 * bound / voxel)
        c = np.arange(lo_c, -lo_c, dtype=np.int64)
        v = (c * voxel).astype(np.float32)                     # (quantized_coords * voxel_size).astype(np.float32)
        inb = (np.float32(-bound) < v) & (v < np.float32(bound))
        t = (v - np.float32(-bound)) / np.float32(grid)
        px = np.floor(t).astype(np.int64)
        py = np.floor(np.float32(S) - t).astype(np.int64) - 1
        px = np.where(px < 0, px + S, px)                      # numpy indexing wraps negatives like torch
        py = np.where(py < 0, py + S, py)
        lut_x = np.where(inb & (px >= 0) & (px < S), px, -1).astype(np.int32)
        lut_y = np.where(inb & (py >= 0) & (py < S), py, -1).astype(np.int32)
        lut_z = ((np.float32(z_range[0]) < v) & (v < np.float32(z_range[1]))).astype(np.int32)
        _LABEL_LUTS[key] = (lut_x, lut_y, lut_z, lo_c, S)
    return _LABEL_LUTS[key]


_DEV_LABEL_LUTS = {}


def bev_labels(coords, labels, bound=50.0, img_size=167, voxel=0.05, batch_size=None):
    """coords int32 [N,4] (batch, x, y, z) on the GPU, labels [N] -> (img_labels int64 [B,S,S], point_idx int32
    [B,S,S]) exactly as getBEVImageNew applied scan by scan (ignore label -1 skipped, last point wins).
    `batch_size`: number of scans in the batch when the caller knows it (a collate function does): nothing is read
    back from the device then."""
    _lib.require_gpu(coords, "coordinates")
    dev = coords.device
    key = (float(bound), int(img_size), float(voxel), str(dev))
    if key not in _DEV_LABEL_LUTS:
        lx, ly, lz, lo, S = label_luts(bound, img_size, voxel)
        _DEV_LABEL_LUTS[key] = (torch.from_numpy(lx).to(dev), torch.from_numpy(ly).to(dev),
                                torch.from_numpy(lz).to(dev), lo, S)
    lx, ly, lz, lo, S = _DEV_LABEL_LUTS[key]
    coords = coords.contiguous()
    lab = labels.to(torch.int32).contiguous()
    B = int(batch_size) if batch_size is not None else int(coords[:, 0].max().item()) + 1
    counts = torch.bincount(coords[:, 0].long(), minlength=B)
    start = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(counts, 0)
    pidx = torch.full((B, S, S), -1, dtype=torch.int32, device=dev)
    img = torch.empty((B, S, S), dtype=torch.int64, device=dev)
    call("lidog_bev_label_raster", ptr(coords), ptr(lab), coords.shape[0], ptr(lx), ptr(ly), ptr(lz), lo, lx.shape[0],
         B, S, ptr(start), ptr(pidx), ptr(img))
    return img, pidx


def mix3d_merge(scan0, scan1, voxel_size=0.05, ignore_label=-1):
    """Mix3DSourceDataset.merge_data (utils/datasets/mix3D.py:44-87): union of two voxelised scans, re-quantised.
    scan = dict(coordinates int [n,3], features [n,C], sem_labels [n]) on the GPU.  As in the reference the
    coordinates go through float32 (`coordinates * voxel_size`, then floor(x / voxel_size)), and the label of a
    merged voxel is the label of its FIRST point (the voted labels returned by sparse_quantize are discarded)."""
    coords = torch.cat([scan0["coordinates"], scan1["coordinates"]], dim=0).float() * voxel_size
    feats = torch.cat([scan0["features"], scan1["features"]], dim=0)
    labels = torch.cat([scan0["sem_labels"], scan1["sem_labels"]], dim=0)
    q, _, _, idx = sparse_quantize(coords, feats, labels=labels, ignore_label=ignore_label,
                                   quantization_size=voxel_size, return_index=True)
    return {"coordinates": q, "features": feats[idx], "sem_labels": labels[idx], "index": idx}
