"""Synthetic LiDAR scans for benchmarks and parity tests (SURVEY.md 8(d) generator).

A spinning sensor at height h above a ground plane, 64 azimuth sectors each closed by a wall at a
random distance; points are the nearer of ground hit and wall hit plus range noise.  Scan i uses
seed i.  Voxelisation follows the dataset code (utils/datasets/semantickitti_bev.py:155-172,187,
232-238): drop points beyond 50 m, apply the bounds filter, floor(p / voxel), keep the first point of
every voxel; features are ones (use_intensity=False, :191-194); labels are uniform in [-1, 6].
"""
import os

import numpy as np
import torch

CONFIGS = {
    # name: beams, azimuths, elevation range (deg), sensor height, voxel size, bounds filter
    "kitti120k": dict(n_beams=64, n_az=1875, elev=(-24.8, 2.0), h=1.73, voxel=0.05, lidog_bounds=True),
    "source8k": dict(n_beams=16, n_az=500, elev=(-24.8, 2.0), h=1.73, voxel=0.1, lidog_bounds=False),
    "nusc35k": dict(n_beams=32, n_az=1090, elev=(-30.0, 10.0), h=1.84, voxel=0.05, lidog_bounds=False),
    "highres524k": dict(n_beams=128, n_az=4096, elev=(-24.8, 2.0), h=1.73, voxel=0.02, lidog_bounds=True),
}


def scan_points(seed, n_beams, n_az, elev, h, **_):
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(elev[0], elev[1], n_beams))
    # evenly spaced azimuths (a spinning sensor fires at a fixed angular step): this is the generator behind the
    # per-stride voxel counts of SURVEY.md 8(d) / BASELINE.md (seed 0, kitti120k: 88 117 / 51 939 / 25 521 / 10 354 /
    # 3 777), asserted by tests/test_oracle_cpu.py::test_synth_matches_baseline_counts and by bench.py
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    wall = rng.uniform(5.0, 50.0, 64)
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    sector = np.minimum((AZ / (2 * np.pi) * 64).astype(np.int64), 63)
    r_wall = wall[sector] / np.cos(EL)
    with np.errstate(divide="ignore"):
        r_ground = np.where(EL < 0, h / np.sin(-EL), np.inf)
    r = np.minimum(r_ground, r_wall) + rng.normal(0.0, 0.02, EL.shape)
    pts = np.stack([r * np.cos(EL) * np.cos(AZ), r * np.cos(EL) * np.sin(AZ), r * np.sin(EL)], axis=-1)
    pts = pts.reshape(-1, 3).astype(np.float32)
    return pts[(pts ** 2).sum(axis=1) < 50.0 ** 2], rng


def voxelize(pts, voxel, lidog_bounds):
    if lidog_bounds:
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        keep = (np.abs(x) < 60) & (np.abs(y) < 60) & (z > -10) & (z < 8) & ~((np.abs(x) < 3) & (np.abs(y) < 2))
        pts = pts[keep]
    vox = np.floor(pts / np.float32(voxel)).astype(np.int32)
    _, first = np.unique(vox, axis=0, return_index=True)
    return vox[np.sort(first)]


# voxels per tensor stride 1/2/4/8/16 of scan seed 0
BASELINE_COUNTS = {
    "kitti120k": (88117, 51939, 25521, 10354, 3777),
    "source8k": (6948, 5033, 3205, 1691, 763),
    "highres524k": (460962, 302486, 144935, 58803, 20581),
    "nusc35k+mix3d": (51943, 37745, 25811, 14387, 6939),
}


def stride_counts(vox):
    """voxels at tensor stride 1, 2, 4, 8, 16 (floor division, as the strided coordinate maps)"""
    return (len(vox),) + tuple(len(np.unique(np.floor_divide(vox, s), axis=0)) for s in (2, 4, 8, 16))


def scan_voxels(seed, config="kitti120k"):
    """(coords int32 [n,3], labels int64 [n]) of synthetic scan `seed`"""
    cfg = CONFIGS[config]
    pts, rng = scan_points(seed, **cfg)
    vox = voxelize(pts, cfg["voxel"], cfg["lidog_bounds"])
    labels = rng.integers(-1, 7, vox.shape[0])
    return vox, labels


def mix3d_voxels(seed, config="nusc35k"):
    """Mix3D-style union of scans 2*seed and 2*seed+1, re-voxelised (utils/datasets/mix3D.py:44-87)"""
    a, la = scan_voxels(2 * seed, config)
    b, lb = scan_voxels(2 * seed + 1, config)
    vox = np.concatenate([a, b])
    lab = np.concatenate([la, lb])
    _, first = np.unique(vox, axis=0, return_index=True)
    first = np.sort(first)
    return vox[first], lab[first]


def make_batch(seeds, config="kitti120k", device="cpu", bev_size=167, mix3d=False):
    """Collated batch with the keys of CollateFNSingleSourceBEVMultiLevel (collation.py:318-325)."""
    coords, labels = [], []
    for b, s in enumerate(seeds):
        v, l = (mix3d_voxels if mix3d else scan_voxels)(s, config)
        coords.append(np.concatenate([np.full((v.shape[0], 1), b, np.int32), v], axis=1))
        labels.append(l)
    coords, labels = np.concatenate(coords), np.concatenate(labels)
    coords = torch.from_numpy(coords).to(device)
    labels = torch.from_numpy(labels).long().to(device)
    rng = np.random.default_rng(1000003 + int(seeds[0]))
    bev = torch.from_numpy(rng.integers(-1, 7, (len(seeds), bev_size, bev_size))).long().to(device)
    feats = torch.ones((coords.shape[0], 1), dtype=torch.float32, device=device)
    return {"source_coordinates0": coords.float(), "source_features0": feats, "source_sem_labels0": labels,
            "source_bev_labels0": {"block8": bev}, "coords_int": coords}
