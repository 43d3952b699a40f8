"""Edge cases of the hot path on the GPU against the CPU oracle: a scan without voxels inside a batch, a ragged batch
(one tiny scan, one normal), duplicate input coordinates through the whole model, inputs of a handful of voxels, an
empty input, and a second forward pass through a manager that has already been used."""
import numpy as np
import pytest
import torch

from helpers import seeded_state_dict, small_scene

pytestmark = pytest.mark.gpu


def _models(kind="MinkUNet34BEV", seed=5):
    import lidog_amd
    import oracle.me_cpu as OME
    from oracle.ref_torch import Encoder2DRef, sparse2super_ref
    from lidog_amd.minkunet import make_models
    kw = dict(in_channels=1, out_channels=7, D=3)
    if kind == "MinkUNet34BEV":
        kw.update(decoder_2d_level=["block8"], mapping_bound_2d=5.0)
    gpu = getattr(lidog_amd, kind)(**kw)
    sd = seeded_state_dict(gpu, seed=seed)
    gpu.load_state_dict(sd)
    OME.set_mode("exact")
    ref_cls = make_models(OME, Encoder2DRef, lambda x, bound, voxel, pool: sparse2super_ref(x.C, x.F, bound, voxel, pool))
    ref = getattr(ref_cls, kind)(**kw)
    ref.load_state_dict(sd)
    return gpu.cuda().train(), ref.train(), OME


def _batch(scans):
    """list of [n_i, 3] int32 arrays (None = a scan without voxels) -> coords [N, 4]"""
    rows = [np.concatenate([np.full((v.shape[0], 1), b, np.int32), v], axis=1) for b, v in enumerate(scans) if v is not None]
    return torch.from_numpy(np.concatenate(rows, axis=0))


def _compare(coords, kind="MinkUNet34BEV", atol=1e-4, train=True):
    import lidog_amd.me as ME
    gpu, ref, OME = _models(kind)
    gpu.train(train), ref.train(train)
    feats = torch.ones((coords.shape[0], 1))
    if kind == "MinkUNet34BEV":
        sg, bg = gpu(ME.SparseTensor(coordinates=coords.cuda(), features=feats.cuda()), is_train=True)
        sr, br = ref(OME.SparseTensor(coordinates=coords, features=feats), is_train=True)
        assert bg["block8"].shape == br["block8"].shape
        assert (bg["block8"].detach().cpu() - br["block8"].detach()).abs().max().item() <= atol
        loss_g = sg.F.square().mean() + bg["block8"].square().mean()
        loss_r = sr.F.square().mean() + br["block8"].square().mean()
    else:
        sg = gpu(ME.SparseTensor(coordinates=coords.cuda(), features=feats.cuda()), is_seg=True)
        sr = ref(OME.SparseTensor(coordinates=coords, features=feats), is_seg=True)
        loss_g, loss_r = sg.F.square().mean(), sr.F.square().mean()
    assert sg.F.shape == sr.F.shape
    assert (sg.F.detach().cpu() - sr.F.detach()).abs().max().item() <= atol
    loss_g.backward()
    loss_r.backward()
    for n in ("final.kernel", "final.bias"):
        a, b = dict(gpu.named_parameters())[n].grad.cpu(), dict(ref.named_parameters())[n].grad
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-8, n
    assert all(torch.isfinite(p.grad).all() for p in gpu.parameters() if p.grad is not None)
    return gpu, sg


def test_batch_with_an_empty_scan_and_a_ragged_batch():
    """batch indices 0 and 2 only (scan 1 has no voxel: its BEV image is all zeros), and a 40-voxel scan next to a
    3 k-voxel one"""
    a, c = small_scene(31, n_points=1500), small_scene(32, n_points=1500)
    coords = _batch([a, None, c])
    gpu, sg = _compare(coords)
    assert sg.coordinate_manager.batch_size == 3
    tiny = small_scene(33, n_points=60, oob=4)
    _compare(_batch([tiny, small_scene(34, n_points=3000)]))


def test_duplicate_input_coordinates_through_the_model():
    """ME keeps the first occurrence of a duplicated coordinate; logits come out per UNIQUE voxel in first-occurrence order"""
    import lidog_amd.me as ME
    v = small_scene(35, n_points=1200)
    dup = np.concatenate([v, v[::7]], axis=0)                       # every 7th voxel twice
    coords = _batch([dup])
    gpu, ref, OME = _models("MinkUNet34")
    feats = torch.arange(coords.shape[0], dtype=torch.float32).view(-1, 1) % 3 + 1
    sg = gpu(ME.SparseTensor(coordinates=coords.cuda(), features=feats.cuda()), is_seg=True)
    sr = ref(OME.SparseTensor(coordinates=coords, features=feats), is_seg=True)
    assert sg.F.shape[0] == v.shape[0] == sr.F.shape[0]
    assert torch.equal(sg.C.cpu(), sr.C)
    assert (sg.F.detach().cpu() - sr.F.detach()).abs().max().item() <= 1e-4


@pytest.mark.parametrize("n", [2, 5, 17])
def test_a_handful_of_voxels(n):
    """maps, tiles and work lists of a few rows (every kernel's tail path), forward and backward.  BatchNorm in
    evaluation mode: batch statistics over one or two rows (the deep strides of such an input) divide rounding noise
    by sqrt(eps) and make any comparison meaningless"""
    g = np.random.default_rng(n)
    v = np.unique(g.integers(-3, 4, (n * 3, 3)).astype(np.int32), axis=0)[:n]
    assert v.shape[0] == n
    _compare(_batch([v]), kind="MinkUNet34", atol=2e-4, train=False)


def test_empty_input_tensor():
    import lidog_amd.me as ME
    coords = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    st = ME.SparseTensor(coordinates=coords, features=torch.zeros((0, 1), device="cuda"))
    conv = ME.MinkowskiConvolution(1, 32, kernel_size=3, dimension=3).cuda()
    out = conv(st)
    assert out.F.shape == (0, 32) and st.coordinate_manager.batch_size == 0
    out2 = ME.MinkowskiConvolution(32, 32, kernel_size=2, stride=2, dimension=3).cuda()(out)
    assert out2.F.shape == (0, 32)


def test_second_forward_pass_on_a_used_manager_drops_the_same_duplicates():
    """a SparseTensor built from (coordinates, manager) AFTER the manager has been handed over must still
    drop the duplicate rows of its feature matrix; a feature matrix of the wrong length raises"""
    import lidog_amd.me as ME
    v = small_scene(36, n_points=800)
    dup = np.concatenate([v, v[:50]], axis=0)
    coords = _batch([dup]).cuda()
    feats = torch.randn(coords.shape[0], 1, device="cuda")
    first = ME.SparseTensor(coordinates=coords, features=feats)
    cm = first.coordinate_manager
    again = ME.SparseTensor(features=feats * 2, coordinates=coords, coordinate_manager=cm)
    assert again.F.shape[0] == v.shape[0] and torch.equal(again.F, first.F * 2)
    with pytest.raises(ValueError):
        ME.SparseTensor(features=feats, coordinate_manager=cm, coordinate_map_key=1)   # 850 rows on an 800-row map
    with pytest.raises(TypeError):
        ME.SparseTensor(coordinates=coords, features=feats, quantization_mode="x")
