"""CPU suite (no GPU): the oracle against dense torch convolutions and against the golden vectors that
the reference's own code produced; host-side logic; the C ABI exports."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from helpers import GOLDEN, REPO, seeded_state_dict, small_batch, sha_triples


def _sha(t):
    return hashlib.sha1(np.ascontiguousarray(t.detach().numpy()).tobytes()).hexdigest()


# ---------------------------------------------------------------- oracle: dense equivalence (G6 of SURVEY 8c)
def _dense_case(seed=0, B=2, S=10, Cin=5):
    torch.manual_seed(seed)
    occ = torch.rand(B, S, S, S) < 0.3
    idx = occ.nonzero().int()
    idx = idx[torch.randperm(idx.shape[0])].contiguous()
    return idx, torch.randn(idx.shape[0], Cin), B, S


def _densify(st, C, B, S, stride=1):
    d = torch.zeros(B, C, S // stride, S // stride, S // stride)
    c = st.C.long()
    d[c[:, 0], :, c[:, 1] // stride, c[:, 2] // stride, c[:, 3] // stride] = st.F.detach()
    return d


@pytest.mark.parametrize("k,s", [(3, 1), (5, 1), (1, 1), (2, 2)])
def test_oracle_conv_equals_dense_conv3d(k, s):
    import oracle.me_cpu as ME
    ME.set_mode("exact")
    idx, F, B, S = _dense_case()
    x = ME.SparseTensor(coordinates=idx, features=F)
    conv = ME.MinkowskiConvolution(5, 7, kernel_size=k, stride=s, dimension=3)
    y = conv(x)
    W = conv.kernel.detach().view(k ** 3, 5, 7)
    Wd = W.view(k, k, k, 5, 7).permute(4, 3, 2, 1, 0).contiguous()  # offsets run x fastest
    yd = Fn.conv3d(_densify(x, 5, B, S), Wd, padding=k // 2 if s == 1 else 0, stride=s)
    c = y.C.long()
    ref = yd[c[:, 0], :, c[:, 1] // s, c[:, 2] // s, c[:, 3] // s]
    assert (y.F - ref).abs().max().item() < 2e-6


def test_oracle_transposed_conv_equals_dense_and_reuses_map():
    import oracle.me_cpu as ME
    ME.set_mode("exact")
    idx, F, B, S = _dense_case(1)
    x = ME.SparseTensor(coordinates=idx, features=F)
    y2 = ME.MinkowskiConvolution(5, 7, kernel_size=2, stride=2, dimension=3)(x)
    tr = ME.MinkowskiConvolutionTranspose(7, 4, kernel_size=2, stride=2, dimension=3)
    z = tr(y2)
    assert torch.equal(z.C, x.C)  # lands on the encoder's map, same row order (needed by ME.cat)
    Wt = tr.kernel.detach().view(2, 2, 2, 7, 4).permute(3, 4, 2, 1, 0).contiguous()
    zd = Fn.conv_transpose3d(_densify(y2, 7, B, S, 2), Wt, stride=2)
    c = z.C.long()
    assert (z.F - zd[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]).abs().max().item() < 2e-6


def test_oracle_gradients_match_autograd_of_dense():
    import oracle.me_cpu as ME
    ME.set_mode("exact")
    idx, F, B, S = _dense_case(2)
    Fg = F.clone().requires_grad_(True)
    conv = ME.MinkowskiConvolution(5, 7, kernel_size=3, dimension=3)
    y = conv(ME.SparseTensor(coordinates=idx, features=Fg))
    g = torch.randn_like(y.F)
    y.F.backward(g)
    # dense twin
    Wd = conv.kernel.detach().view(3, 3, 3, 5, 7).permute(4, 3, 2, 1, 0).contiguous().requires_grad_(True)
    c = idx.long()
    Fd = F.clone().requires_grad_(True)
    d = torch.zeros(B, S, S, S, 5).index_put((c[:, 0], c[:, 1], c[:, 2], c[:, 3]), Fd).permute(0, 4, 1, 2, 3)
    yd = Fn.conv3d(d, Wd, padding=1)[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    yd.backward(g)
    assert (Fg.grad - Fd.grad).abs().max().item() < 1e-5
    gw = Wd.grad.permute(4, 3, 2, 1, 0).reshape(27, 5, 7)
    assert (conv.kernel.grad - gw).abs().max().item() < 1e-4
    # blas mode (the timed CPU baseline) agrees with the exact mode
    ME.set_mode("blas")
    Fb = F.clone().requires_grad_(True)
    conv.kernel.grad = None
    yb = conv(ME.SparseTensor(coordinates=idx, features=Fb))
    yb.F.backward(g)
    ME.set_mode("exact")
    assert (yb.F - y.F).abs().max().item() < 1e-5 and (Fb.grad - Fg.grad).abs().max().item() < 1e-5


def test_oracle_stride_floor_and_first_occurrence():
    import oracle.me_cpu as ME
    c = torch.tensor([[0, -1, 0, 3], [0, 1, 1, 2], [0, -2, -1, 2], [1, -1, 0, 3], [0, -3, 0, 0]], dtype=torch.int32)
    st = ME.SparseTensor(coordinates=c, features=torch.ones(5, 1))
    out = st.coordinate_manager.stride(1, 2)
    # floor toward -inf: -1 -> -2, -3 -> -4; rows in first-occurrence order, batch column untouched
    assert out.tolist() == [[0, -2, 0, 2], [0, 0, 0, 2], [0, -2, -2, 2], [1, -2, 0, 2], [0, -4, 0, 0]]


def test_sparse_quantize_arity_and_labels():
    import oracle.me_cpu as ME
    pts = np.array([[0.01, 0.01, 0.0], [0.02, 0.03, 0.01], [0.11, 0.0, 0.0], [0.12, 0.01, 0.0]], np.float32)
    lab = np.array([3, 4, 2, 2])
    q = ME.utils.sparse_quantize(pts, pts, labels=lab, ignore_label=-1, quantization_size=0.05,
                                 return_index=True, return_inverse=True)
    assert len(q) == 5
    assert q[0].tolist() == [[0, 0, 0], [2, 0, 0]] and q[2].tolist() == [-1, 2]
    assert q[3].tolist() == [0, 2] and q[4].tolist() == [0, 0, 1, 1]
    assert len(ME.utils.sparse_quantize(pts, quantization_size=[0.05, 0.05, 0.05], return_index=True,
                                        return_inverse=True)) == 3


# ---------------------------------------------------------------- restatements vs reference-made golden vectors
def test_sparse2super_restatement_bit_equals_reference():
    from oracle.ref_torch import sparse2super_ref
    g2 = np.load(f"{GOLDEN}/g2_sparse2super.npz")
    C = torch.from_numpy(g2["coords"])
    g = torch.Generator().manual_seed(int(g2["seed"]))
    F = torch.rand((C.shape[0], 96), generator=g)
    F = torch.where(torch.rand(F.shape, generator=g) < 0.3, torch.zeros_like(F), F).requires_grad_(True)
    out = sparse2super_ref(C, F, 5.0)
    assert _sha(out) == str(g2["out_sha1"])
    out.backward(torch.randn(out.shape, generator=g))
    torch.testing.assert_close(F.grad.sum(dim=1), torch.from_numpy(g2["gin_rowsum"]), rtol=1e-5, atol=1e-5)


def test_encoder2d_restatement_equals_reference():
    from oracle.ref_torch import Encoder2DRef
    g3 = np.load(f"{GOLDEN}/g3_encoder2d.npz")
    enc = Encoder2DRef(96, 7)
    assert list(enc.state_dict().keys()) == list(g3["keys"])
    enc.load_state_dict(seeded_state_dict(enc, seed=3))
    enc.train()
    g = torch.Generator().manual_seed(13)
    x = torch.rand((2, 96, 66, 66), generator=g)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.8, torch.zeros_like(x), x)
    torch.testing.assert_close(enc(x), torch.from_numpy(g3["y"]), rtol=1e-5, atol=1e-5)


def test_loss_restatements_equal_reference():
    from oracle.ref_torch import soft_dice_loss_ref, dice_loss_ref
    from lidog_amd.losses import SoftDICELoss, DICELoss
    g4 = np.load(f"{GOLDEN}/g4_losses.npz")
    logits, labels = torch.from_numpy(g4["logits"]), torch.from_numpy(g4["labels"])
    assert abs(float(soft_dice_loss_ref(logits, labels)) - float(g4["soft_dice"])) < 1e-7
    assert abs(float(dice_loss_ref(logits, labels)) - float(g4["dice"])) < 1e-7
    lg = logits.clone().requires_grad_(True)
    l = soft_dice_loss_ref(lg, labels)
    l.backward()
    torch.testing.assert_close(lg.grad, torch.from_numpy(g4["soft_dice_grad"]), rtol=1e-4, atol=1e-8)
    bev, bl = torch.from_numpy(g4["bev"]), torch.from_numpy(g4["bev_labels"])
    assert abs(float(dice_loss_ref(bev.view(-1, 7), bl.view(-1))) - float(g4["bev_dice"])) < 1e-6
    # the product losses are HIP kernels only: CPU tensors are refused (tests/test_gpu_bev_head.py checks them on the GPU)
    for crit in (SoftDICELoss(ignore_label=-1), DICELoss(ignore_label=-1)):
        with pytest.raises(RuntimeError):
            crit(logits, labels)


def test_wiring_on_oracle_equals_reference_model_golden():
    """lidog_amd.minkunet bound to the CPU oracle reproduces what the REFERENCE class produced (G6)."""
    import oracle.me_cpu as OME
    from lidog_amd.minkunet import make_models
    from oracle.ref_torch import soft_dice_loss_ref
    OME.set_mode("blas")
    try:
        g6 = np.load(f"{GOLDEN}/g6_minkunet34.npz")
        C, labels = torch.from_numpy(g6["coords"]), torch.from_numpy(g6["labels"])
        model = make_models(OME).MinkUNet34(in_channels=1, out_channels=7, D=3)
        assert list(model.state_dict().keys()) == list(g6["keys"])
        model.load_state_dict(seeded_state_dict(model, seed=7))
        model.train()
        sem = model(OME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1))))
        assert (sem.F.detach() - torch.from_numpy(g6["logits"])).abs().max().item() < 5e-5
        assert abs(float(soft_dice_loss_ref(sem.F, labels)) - float(g6["loss"])) < 1e-5
    finally:
        OME.set_mode("exact")


def test_golden_kernel_maps_reproduce():
    import oracle.me_cpu as OME
    g5 = np.load(f"{GOLDEN}/g5_minkunet34bev.npz")
    C = torch.from_numpy(g5["coords"])
    st = OME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1)))
    cm = st.coordinate_manager
    prev = 1
    for s in (2, 4, 8, 16):
        cm.stride(prev, s)
        prev = s
    assert [cm.maps[s].shape[0] for s in (1, 2, 4, 8, 16)] == g5["n_vox"].tolist()
    k_off, pin, pout, _ = cm.kernel_map(1, 1, 3)
    assert sha_triples(k_off, pin, pout) == str(g5["kmap_1_1_3"][0])


# ---------------------------------------------------------------- host logic of the product (no kernels run)
def test_pixel_luts_match_reference_golden():
    from lidog_amd.bev import pixel_luts, bev_image_size
    g1 = np.load(f"{GOLDEN}/g1_bev_luts.npz")
    for b in (50, 30, 5):
        lx, ly, lo, H = pixel_luts(float(b))
        assert H == bev_image_size(float(b)) == {50: 2000, 30: 1200, 5: 200}[b]
        assert np.array_equal(lx, g1[f"lut_x_{b}"]) and np.array_equal(ly, g1[f"lut_y_{b}"])
    lx, _, lo, _ = pixel_luts(50.0)
    naive = np.arange(lo, lo + lx.shape[0]) + 1000
    off = int(((lx >= 0) & (lx != naive)).sum())
    assert off > 100  # the float32 rounding quirk the reference bakes into the trained model


def test_tile_descriptors_and_offsets():
    from lidog_amd.me import _tiles, kernel_offsets
    desc, n = _tiles([0, 5, 5, 300, 428], "cpu")
    assert n == 5
    # 128-row tiles that never straddle an offset, launched in order of relative position inside their segment
    assert desc.tolist() == [[2, 0, 2, 3, 2], [5, 0, 133, 300, 261], [128, 5, 128, 128, 39]]
    o3 = kernel_offsets(3, 2)
    assert o3.shape == (27, 3) and o3[0].tolist() == [-2, -2, -2] and o3[1].tolist() == [0, -2, -2]
    assert o3[13].tolist() == [0, 0, 0]
    o2 = kernel_offsets(2, 4)
    assert o2.tolist() == [[0, 0, 0], [4, 0, 0], [0, 4, 0], [4, 4, 0], [0, 0, 4], [4, 0, 4], [0, 4, 4], [4, 4, 4]]


def test_model_parameter_count_and_names():
    import lidog_amd
    m = lidog_amd.MinkUNet34BEV(1, 7, 3)
    assert sum(p.numel() for p in m.parameters()) == 38660622
    g5 = np.load(f"{GOLDEN}/g5_minkunet34bev.npz")
    assert list(m.state_dict().keys()) == list(g5["keys"])
    convs = [x for x in m.modules() if isinstance(x, (lidog_amd.me.MinkowskiConvolution,
                                                      lidog_amd.me.MinkowskiConvolutionTranspose))]
    bns = [x for x in m.modules() if isinstance(x, lidog_amd.me.MinkowskiBatchNorm)]
    assert len(convs) == 63 and len(bns) == 62


def test_product_refuses_cpu_tensors():
    import lidog_amd.me as ME
    with pytest.raises(RuntimeError):
        ME.SparseTensor(coordinates=torch.zeros((4, 4), dtype=torch.int32), features=torch.ones(4, 1))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "lidog_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


# ---------------------------------------------------------------- the C ABI
def test_shared_library_exports_every_declared_symbol():
    from lidog_amd import build, _lib
    so = build.build()
    lib = ctypes.CDLL(so)
    header = open(os.path.join(REPO, "include", "lidog_amd.h")).read()
    declared = set(re.findall(r"\b(lidog_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 27
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lidog_amd.h but not exported"
    assert declared - {"lidog_last_error"} == set(_lib.SIGNATURES), "python binding and header disagree"
    assert lib.lidog_abi_version() == _lib.ABI_VERSION == 8
    lib.lidog_hash_capacity.restype = ctypes.c_int64
    assert lib.lidog_hash_capacity(ctypes.c_int64(1000)) == 2048


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils/models"), reason="reference only exists in the build container")
def test_reference_model_file_builds_on_product_api():
    """INTEGRATION.md route 1: the reference's own minkunet_bev.py constructs on top of lidog_amd.me
    (run in a subprocess so the alias does not leak into this interpreter)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.dont_write_bytecode = True; sys.path.insert(0, %r); sys.path.insert(1, '/root/reference');"
        "import lidog_amd.me as ME; ME.install_as_minkowski_engine();"
        "from utils.models.minkunet_bev import MinkUNet34BEV as Ref; import lidog_amd;"
        "import inspect, MinkowskiEngine as M;"
        "sig = inspect.signature(M.utils.sparse_quantize).parameters;"
        "assert list(sig)[:4] == ['coordinates', 'features', 'labels', 'ignore_label'] and "
        "{'return_index', 'return_inverse', 'quantization_size'} <= set(sig);"
        "r = Ref(in_channels=1, out_channels=7, D=3); m = lidog_amd.MinkUNet34BEV(1, 7, 3);"
        "assert list(r.state_dict().keys()) == list(m.state_dict().keys());"
        "assert all(a.shape == b.shape for a, b in zip(r.state_dict().values(), m.state_dict().values()));"
        "print('ok')" % REPO)
    out = subprocess.run([sys.executable, "-B", "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_bev_label_luts_reproduce_reference_golden():
    """host arithmetic of lidog_amd.data.label_luts against images made by the reference's getBEVImageNew"""
    from lidog_amd.data import label_luts
    g7 = np.load(f"{GOLDEN}/g7_bev_labels.npz")
    for bound, size in ((50.0, 167), (30.0, 100)):
        tag = str(int(bound))
        vox, lab = g7[f"vox_{tag}"], g7[f"labels_{tag}"]
        lx, ly, lz, lo, S = label_luts(bound, size, 0.05)
        assert S == size
        j = vox - lo
        ok = (lab != -1) & (j >= 0).all(1) & (j < lx.shape[0]).all(1)
        jj = np.clip(j, 0, lx.shape[0] - 1)
        px, py, pz = lx[jj[:, 0]], ly[jj[:, 1]], lz[jj[:, 2]]
        ok &= (px >= 0) & (py >= 0) & (pz == 1)
        img = -np.ones((S, S), np.int32)
        idx = -np.ones((S, S), np.int32)
        rows = np.nonzero(ok)[0]
        img[py[rows], px[rows]] = lab[rows]   # numpy assignment order: the last row wins, as in the reference
        idx[py[rows], px[rows]] = rows
        assert np.array_equal(img, g7[f"img_{tag}"]) and np.array_equal(idx, g7[f"idx_{tag}"])


def test_iou_definition_matches_sklearn_as_the_reference_calls_it():
    """trainer_lighting_bev.py:282-293: jaccard_score(preds, labels, average=None, labels=arange(C), zero_division=0)
    with absent classes set to -1, then nan-mean aggregation (:356-370)"""
    from sklearn.metrics import jaccard_score
    from lidog_amd.evaluate import per_class_iou, mean_iou
    g = torch.Generator().manual_seed(3)
    rows, ref_rows = [], []
    for n in (500, 37, 1200):
        preds = torch.randint(0, 7, (n,), generator=g)
        labels = torch.randint(-1, 5 if n == 37 else 7, (n,), generator=g)   # one scan without classes 5, 6
        ref = jaccard_score(preds.numpy(), labels.numpy(), average=None, labels=np.arange(7), zero_division=0.)
        present = np.unique(labels.numpy())
        present = present[present != -1]
        r = -np.ones(7)
        r[present] = ref[present]
        got = per_class_iou(preds, labels, 7, -1)
        assert np.allclose(got.numpy(), r, atol=1e-12)
        rows.append(got)
        ref_rows.append(r)
    per_class, mean = mean_iou(torch.stack(rows))
    m = np.stack(ref_rows)
    m[m == -1] = np.nan
    assert np.allclose(per_class.numpy(), np.nanmean(m, axis=0) * 100) and abs(float(mean) - np.nanmean(np.nanmean(m, axis=0) * 100)) < 1e-9


def test_lightning_checkpoint_round_trip(tmp_path):
    import lidog_amd
    from lidog_amd.checkpoint import load_lightning_checkpoint, save_lightning_checkpoint, model_state_dict
    m = lidog_amd.MinkUNet34(1, 7, 3)
    sd = seeded_state_dict(m, seed=9)
    m.load_state_dict(sd)
    path = str(tmp_path / "epoch=4-step=100.ckpt")
    save_lightning_checkpoint(m, path, epoch=4, global_step=100)
    raw = torch.load(path, weights_only=False)
    assert all(k.startswith("model.") for k in raw["state_dict"])           # the key layout Lightning writes
    assert "model.block1.0.conv1.kernel" in raw["state_dict"] and "model.bn0.bn.running_var" in raw["state_dict"]
    m2 = lidog_amd.MinkUNet34(1, 7, 3)
    epoch, _ = load_lightning_checkpoint(m2, path)
    assert epoch == 4 and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert set(model_state_dict(raw)) == set(sd)


def test_synth_matches_baseline_counts():
    """the bench workload IS BASELINE.md's: per-stride voxel counts of seed 0 equal SURVEY.md 8(d)'s probe numbers"""
    from lidog_amd import synth
    for cfg in ("kitti120k", "source8k"):
        vox, _ = synth.scan_voxels(0, cfg)
        assert synth.stride_counts(vox) == synth.BASELINE_COUNTS[cfg], cfg
    vox, _ = synth.mix3d_voxels(0)
    assert synth.stride_counts(vox) == synth.BASELINE_COUNTS["nusc35k+mix3d"]
    pts, _ = synth.scan_points(0, **synth.CONFIGS["kitti120k"])
    assert pts.shape[0] == 120000


def test_results_csv_has_the_reference_format(tmp_path):
    """test_epoch_end's CSV (trainer_lighting_bev.py:325-383): header once, decimal commas, nan-mean over scans x 100"""
    import csv
    from lidog_amd.evaluate import write_results_csv
    iou = torch.tensor([[0.5, -1.0, 0.25], [0.7, 0.1, -1.0]], dtype=torch.float64)
    names = ["vehicle", "person", "road"]
    p = write_results_csv(str(tmp_path), "SemanticKITTI", "NuScenes", iou, names)
    write_results_csv(str(tmp_path), "SemanticKITTI", "SemanticPOSS", iou * 0 + 0.123456, names, first_target=False)
    rows = list(csv.reader(open(p)))
    assert p.endswith("results/SemanticKITTI-TO-NuScenes.csv")
    assert rows[0] == ["source", "target", "vehicle", "person", "road", "mean"]
    assert rows[1] == ["SemanticKITTI", "NuScenes", "60,0", "10,0", "25,0", "31,67"]
    other = list(csv.reader(open(str(tmp_path / "results" / "SemanticKITTI-TO-SemanticPOSS.csv"))))
    assert other == [["SemanticKITTI", "SemanticPOSS", "12,35", "12,35", "12,35", "12,35"]]
