"""Coordinate maps built ahead of time on the side stream (CoordinateManager.prepare) must be the maps the lazy
path builds, and a training run that prefetches must follow the same trajectory as one that does not."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches():
    from lidog_amd import synth
    return [synth.make_batch(range(2 * i, 2 * i + 2), "source8k", "cuda") for i in range(2)]


def test_prepared_manager_equals_lazy_manager():
    import lidog_amd
    import lidog_amd.me as ME
    b = _batches()[0]
    model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    lazy = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
    model(lazy).F.sum().backward()
    cm0 = lazy.coordinate_manager
    assert len(cm0.trace) == 63                                    # every convolution of MinkUNet34
    cm1 = ME.CoordinateManager.prepare(b["coords_int"], cm0.trace)
    n_maps = len(cm1.kmaps)
    assert set(cm1.kmaps) == set(cm0.kmaps) and set(cm1.identity) == set(cm0.identity)
    st = ME.SparseTensor(features=b["source_features0"], coordinates=b["coords_int"], coordinate_manager=cm1)
    torch.cuda.synchronize()
    for key, m0 in cm0.kmaps.items():
        m1 = cm1.kmaps[key]
        assert m0.k_off_host == m1.k_off_host and m0.n_tiles == m1.n_tiles
        for name in ("pair_in", "pair_out", "pos_out", "pos_in", "tiles"):
            assert torch.equal(getattr(m0, name), getattr(m1, name)), (key, name)
        assert set(m0._wgrad_items) == set(m1._wgrad_items)
        for chunk, (items, n, off) in m0._wgrad_items.items():
            assert n == m1._wgrad_items[chunk][1]
            assert torch.equal(items, m1._wgrad_items[chunk][0]) and torch.equal(off, m1._wgrad_items[chunk][2])
    for s in cm0.maps:
        assert torch.equal(cm0.maps[s].coords, cm1.maps[s].coords)
    out = model(st)
    out.F.sum().backward()
    assert len(cm1.kmaps) == n_maps, "the forward pass had to build a map the trace did not name"
    torch.testing.assert_close(out.F, model(ME.SparseTensor(coordinates=b["coords_int"],
                                                            features=b["source_features0"])).F, rtol=0, atol=0)


def test_prepare_with_duplicate_coordinates():
    import lidog_amd.me as ME
    b = _batches()[0]
    C = torch.cat([b["coords_int"], b["coords_int"][:100]])
    F = torch.randn(C.shape[0], 4, device="cuda")
    conv = ME.MinkowskiConvolution(4, 8, kernel_size=3, dimension=3).cuda()
    a = ME.SparseTensor(coordinates=C, features=F)
    ya = conv(a)
    cm = ME.CoordinateManager.prepare(C, a.coordinate_manager.trace)
    yb = conv(ME.SparseTensor(features=F, coordinates=C, coordinate_manager=cm))
    assert ya.F.shape[0] == b["coords_int"].shape[0]
    assert torch.equal(ya.F, yb.F)


def test_training_with_prefetch_follows_the_same_trajectory():
    import lidog_amd
    from lidog_amd.trainer import FlatAdam, SourceStep
    batches = _batches()
    torch.manual_seed(3)
    m0 = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    m1 = copy.deepcopy(m0)
    s0 = SourceStep(m0, FlatAdam(m0, lr=1e-3, weight_decay=1e-4))
    s1 = SourceStep(m1, FlatAdam(m1, lr=1e-3, weight_decay=1e-4))
    ready = torch.cuda.Event()
    ready.record()
    torch.cuda.synchronize()
    l0 = [float(s0.training_step(batches[i % 2])["loss"]) for i in range(4)]
    l1 = [float(s1.training_step(batches[i % 2], prefetch=batches[(i + 1) % 2], prefetch_ready=ready)["loss"])
          for i in range(4)]
    assert l0 == l1, (l0, l1)                 # MinkUNet34 has no atomics: bit-identical
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("mode", [1, 2])
def test_second_backward_stream_does_not_change_the_trajectory(mode, monkeypatch):
    """Weight gradients on the second stream (forked before / behind the data gradient's GEMM, joined by the engine
    callback at the end of backward) against everything on one stream: the kernels and their inputs are the same,
    so losses and parameters must be BIT-identical after four Adam steps -- any missing dependency between the two
    streams would show up here (MinkUNet34 has no atomics)."""
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd.trainer import FlatAdam, SourceStep
    batches = _batches()
    monkeypatch.setattr(ME, "_WGRAD_FIT", 0)   # same work items on both sides (the default cuts them by stream mode)
    torch.manual_seed(5)
    m0 = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    m1 = copy.deepcopy(m0)
    s0 = SourceStep(m0, FlatAdam(m0, lr=1e-3, weight_decay=1e-4))
    s1 = SourceStep(m1, FlatAdam(m1, lr=1e-3, weight_decay=1e-4))
    try:
        ME.set_backward_overlap(False)
        l0 = [float(s0.training_step(batches[i % 2])["loss"]) for i in range(4)]
        ME.set_backward_overlap(True, mode)
        l1 = [float(s1.training_step(batches[i % 2])["loss"]) for i in range(4)]
    finally:
        ME.set_backward_overlap(True, 2)
    assert l0 == l1, (l0, l1)
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k
