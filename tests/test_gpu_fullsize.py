"""Full-size checks on the GPU (BASELINE.json configs, where the CPU oracle would take minutes): size-independent
properties of the coordinate maps and of the convolution, run-to-run bit reproducibility, and the training
steps of configs 1, 2, 4 and the forward of config 5 (461 k-voxel stress case)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tensor(config, seeds, mix3d=False):
    import lidog_amd.me as ME
    from lidog_amd import synth
    b = synth.make_batch(seeds, config, "cuda", mix3d=mix3d)
    return ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]), b


def test_maps_at_120k_points():
    st, _ = _tensor("kitti120k", [0, 1])
    cm = st.coordinate_manager
    prev, n_prev = 1, cm.maps[1].n
    for s in (2, 4, 8, 16):
        m = cm.stride(prev, s)
        c = m.coords
        assert m.n < n_prev and torch.all(c[:, 1:] % s == 0)                      # on the coarser lattice
        assert torch.unique(c, dim=0).shape[0] == m.n                              # no duplicates
        # every parent floors onto a row of the child map, and first-occurrence order is monotone
        par = cm.maps[prev].coords.clone()
        par[:, 1:] = torch.div(par[:, 1:], s, rounding_mode="floor") * s
        km = cm.kernel_map(prev, s, 2)                                             # k2 s2: exactly one pair per parent
        assert km.P == n_prev and torch.equal(torch.sort(km.pair_in)[0].long(), torch.arange(n_prev, device="cuda"))
        assert torch.equal(c[km.pair_out.long()][:, 0], par[km.pair_in.long()][:, 0])
        prev, n_prev = s, m.n
    # 3^3 map: symmetric under (k, in, out) <-> (26 - k, out, in); the centre offset is the identity
    km = cm.kernel_map(1, 1, 3)
    K = km.K
    off = torch.tensor(km.k_off_host, device="cuda")
    ks = torch.repeat_interleave(torch.arange(K, device="cuda"), off[1:] - off[:-1])
    n = cm.maps[1].n
    key = (ks * n + km.pair_in.long()) * n + km.pair_out.long()
    mirrored = ((K - 1 - ks) * n + km.pair_out.long()) * n + km.pair_in.long()
    assert torch.equal(torch.sort(key)[0], torch.sort(mirrored)[0])
    a, b = km.k_off_host[13], km.k_off_host[14]
    assert b - a == n and torch.equal(km.pair_in[a:b], km.pair_out[a:b])


def test_conv_linearity_adjointness_and_reproducibility_at_120k_points():
    import lidog_amd.me as ME
    torch.manual_seed(0)   # the kernel initialisation draws from the global generator
    st, _ = _tensor("kitti120k", [2])
    cm, n = st.coordinate_manager, st.F.shape[0]
    g = torch.Generator(device="cuda").manual_seed(0)
    conv = ME.MinkowskiConvolution(96, 96, kernel_size=3, dimension=3).cuda()
    x = torch.randn(n, 96, device="cuda", generator=g, requires_grad=True)
    y = torch.randn(n, 96, device="cuda", generator=g)
    gy = torch.randn(n, 96, device="cuda", generator=g)
    f = lambda t: conv(ME.SparseTensor(t, coordinate_manager=cm, coordinate_map_key=1)).F
    out = f(x)
    lin = f(2.0 * x.detach() - 0.5 * y)
    ref = 2.0 * out.detach() - 0.5 * f(y)
    assert (lin - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    out.backward(gy)
    # <conv(x), g> == <x, conv^T(g)>  (the data gradient is the adjoint)   and   == <W, dW>
    # the inner product is a sum of ~7 M signed terms that cancel almost completely, so the tolerance is
    # relative to the sum of their magnitudes (fp32 rounding of each term), not to the small total
    lhs = (out.detach().double() * gy.double()).sum()
    scale = float((out.detach().double().abs() * gy.double().abs()).sum())
    assert abs(float(lhs - (x.detach().double() * x.grad.double()).sum())) <= 1e-6 * scale
    assert abs(float(lhs - (conv.kernel.detach().double() * conv.kernel.grad.double()).sum())) <= 1e-6 * scale
    # no atomics in the convolution: forward, data gradient and weight gradient are bit-reproducible
    gx1, gw1 = x.grad.clone(), conv.kernel.grad.clone()
    x.grad, conv.kernel.grad = None, None
    out2 = f(x)
    out2.backward(gy)
    assert torch.equal(out2, out) and torch.equal(x.grad, gx1) and torch.equal(conv.kernel.grad, gw1)


def test_bev_projection_properties_at_full_image():
    """B = 50 m (2000 x 2000 px): gradient mass is conserved through pool + scatter for non-colliding voxels"""
    import lidog_amd.me as ME
    from lidog_amd.bev import sparse2super
    st, _ = _tensor("kitti120k", [3, 4])
    n = st.F.shape[0]
    feats = torch.rand(n, 96, device="cuda").requires_grad_(True)
    x = ME.SparseTensor(feats, coordinate_manager=st.coordinate_manager, coordinate_map_key=1)
    out = sparse2super(x, bound=50.0)
    assert tuple(out.shape) == (2, 96, 666, 666) and float(out.min()) >= 0.0
    assert float(out.max()) == float(feats.max())            # max-pool of scattered values: the global max survives
    gout = torch.rand_like(out)
    out.backward(gout)
    assert torch.isfinite(feats.grad).all() and float(feats.grad.sum()) > 0
    out2 = sparse2super(x, bound=50.0)
    assert torch.equal(out, out2)                             # atomicMax winner map: deterministic


def test_first_bev_convolution_over_the_support_of_a_real_scan():
    """B = 50 m image of two 120 k-point scans (95 % empty): the convolution restricted to the active channels of
    each pixel tile (row bitmasks written by the pooling kernel) against the dense kernels on the same image --
    forward bit-identical, data gradient bit-identical wherever the pooling backward reads it, weight gradient
    equal up to the order of the split sums; and the pooling backward through the bitmasks against the dense walk."""
    import lidog_amd.me as ME
    from lidog_amd import bev
    from lidog_amd._lib import call, load, ptr
    st, _ = _tensor("kitti120k", [5, 6])
    n = st.F.shape[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    feats = torch.rand(n, 96, device="cuda", generator=g).requires_grad_(True)
    img = bev.sparse2super(ME.SparseTensor(feats, coordinate_manager=st.coordinate_manager, coordinate_map_key=1), 50.0)
    act = bev.structural_support(img)
    assert act is not None
    B, Cin, H, W = img.shape
    Cout = 256
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
    x = img.detach()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    L = load()
    call("lidog_conv2d_support", None, B, Cin, H, W, ptr(act))
    ws = torch.empty(9 * Cin * Cout, device="cuda")
    y_d, y_s = torch.empty(B, Cout, Ho, Wo, device="cuda"), torch.empty(B, Cout, Ho, Wo, device="cuda")
    call("lidog_conv2d_fwd", ptr(x), ptr(w), None, B, Cin, H, W, Cout, 3, 2, 1, ptr(y_d))
    call("lidog_conv2d_fwd_sparse", ptr(x), ptr(w), ptr(act), B, Cin, H, W, Cout, ptr(y_s), ptr(ws))
    assert torch.equal(y_s, y_d)
    gy = torch.randn(B, Cout, Ho, Wo, device="cuda", generator=g)
    gx_d, gx_s = torch.empty_like(x), torch.zeros_like(x)
    call("lidog_conv2d_dgrad", ptr(gy), ptr(w), B, Cin, H, W, Cout, 3, 2, 1, ptr(gx_d), ptr(ws))
    call("lidog_conv2d_dgrad_sparse", ptr(gy), ptr(w), ptr(act), B, Cin, H, W, Cout, ptr(gx_s), ptr(ws))
    words = (W + 63) // 64
    bits = act[:2 * B * Cin * H * words].view(torch.int64).view(B, Cin, H, words)
    cols = torch.arange(W, device="cuda")
    need = ((bits[..., cols // 64] >> (cols % 64)) & 1).bool()          # the windows the pooling kernel computed
    assert 0.02 < float(need.float().mean()) < 0.10
    assert torch.equal(x != 0, (x != 0) & need)                          # the image is zero outside its support
    assert torch.equal(gx_s[need], gx_d[need])
    gw_d, gw_s = torch.empty_like(w), torch.empty_like(w)
    wsw = torch.empty(max(32 * w.numel(), L.lidog_conv2d_wgrad_sparse_ws(B, Cin, H, W, Cout)), device="cuda")
    call("lidog_conv2d_wgrad", ptr(x), ptr(gy), B, Cin, H, W, Cout, 3, 2, 1, ptr(gw_d), None, ptr(wsw), wsw.numel())
    call("lidog_conv2d_wgrad_sparse", ptr(x), ptr(gy), ptr(act), B, Cin, H, W, Cout, ptr(gw_s), ptr(wsw), wsw.numel())
    assert float((gw_s - gw_d).abs().max()) <= 1e-5 * float(gw_d.abs().max())
    # pooling backward (gather form, no atomics): equal to an index_add over the source map, and run-to-run identical
    gout = torch.randn_like(img)
    winner, pixel, argsrc, _ = img.grad_fn.saved_tensors
    img.backward(gout)
    src = torch.where(need, argsrc, torch.full_like(argsrc, -1))         # argsrc is only defined on the support
    ok = src >= 0
    gcell = torch.zeros(n * 96, device="cuda", dtype=torch.float64)
    gcell.index_add_(0, src[ok].long(), gout[ok].double())
    pix = pixel.long()
    ref = torch.where((pix >= 0).unsqueeze(1), gcell.view(n, 96)[winner.view(-1)[pix.clamp(min=0)].long().clamp(min=0)],
                      torch.zeros((), device="cuda", dtype=torch.float64))
    torch.testing.assert_close(feats.grad.double(), ref, rtol=1e-5, atol=1e-6)   # fp32 sums of <= 4 window gradients
    gfe = torch.empty(n, 96, device="cuda")
    Bp, _, Hp, Wp = winner.shape[0], None, winner.shape[1], winner.shape[2]
    call("lidog_bev_pool_bwd", ptr(gout), ptr(argsrc), ptr(winner), ptr(pixel), n, 96, Bp, Hp, Wp, 5, 3, 1, H, W, ptr(gfe))
    assert torch.equal(gfe, feats.grad)                                  # bit-identical from run to run


@pytest.mark.parametrize("name", ["c1_source8k", "c4_mix3d"])
def test_source_training_configs_learn(name):
    """configs[0] (train_source MinkUNet34, 8k points, 0.1 m voxels, bs 4) and configs[3] (Mix3D union of two
    nuScenes-like scans): a few Adam steps on a fixed batch must reduce the SoftDICE loss"""
    import lidog_amd
    from lidog_amd import synth
    from lidog_amd.trainer import FlatAdam, SourceStep
    torch.manual_seed(0)
    model = lidog_amd.MinkUNet34(1, 7, 3).cuda().train()
    step = SourceStep(model, FlatAdam(model, lr=1e-2, weight_decay=1e-4))
    if name == "c1_source8k":
        batch = synth.make_batch(range(4), "source8k", "cuda")
    else:
        batch = synth.make_batch(range(2), "nusc35k", "cuda", mix3d=True)
    losses = [float(step.training_step(batch)["loss"]) for _ in range(8)]
    # labels are uniform random per voxel (hard to fit): the loss only has to move down from its start
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0] - 0.005, losses


def test_lidog_training_config_learns():
    """configs[1]: MinkUNet34BEV + BEV head at B = 50 m, bs 2 here"""
    import lidog_amd
    from lidog_amd import synth
    from lidog_amd.trainer import FlatAdam, LiDOGStep
    torch.manual_seed(0)
    model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
    step = LiDOGStep(model, FlatAdam(model, lr=1e-3, weight_decay=1e-4))
    batch = synth.make_batch(range(2), "kitti120k", "cuda")
    losses = [float(step.training_step(batch)["loss"]) for _ in range(5)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_highres_stress_forward():
    """configs[4]: 524 288 points at 0.02 m voxels (~420 k active voxels), bs 1: maximum sizes run end to end"""
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd import synth
    torch.manual_seed(0)
    model = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=50.0).cuda().train()
    b = synth.make_batch([0], "highres524k", "cuda")
    assert b["coords_int"].shape[0] > 400000
    sem, bev = model(ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]), is_train=True)
    assert sem.F.shape == (b["coords_int"].shape[0], 7) and tuple(bev["block8"].shape) == (1, 7, 167, 167)
    assert torch.isfinite(sem.F).all() and torch.isfinite(bev["block8"]).all()
    (sem.F.square().mean() + bev["block8"].square().mean()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_bev_head_gradients_are_run_to_run_identical():
    """Encoder2D on the B = 50 m image of two 120 k-point scans, twice: logits, every parameter gradient, the feature
    gradient and the running statistics must be torch.equal between the runs -- no sum on the path depends on the order
    in which workgroups finish (round 3: the BatchNorm2d plane sums are per-(image, chunk) partials added in a fixed
    order; they were fp64 atomics in round 2)"""
    import lidog_amd.me as ME
    from lidog_amd import bev
    st, _ = _tensor("kitti120k", [7, 8])
    n = st.F.shape[0]
    torch.manual_seed(5)
    enc = bev.Encoder2D(96, 7).cuda().train()
    sd0 = {k: v.clone() for k, v in enc.state_dict().items()}
    g = torch.Generator(device="cuda").manual_seed(2)
    base = torch.rand(n, 96, device="cuda", generator=g)
    gl = None
    runs = []
    for _ in range(2):
        enc.load_state_dict(sd0)
        enc.zero_grad()
        feats = base.clone().requires_grad_(True)
        img = bev.sparse2super(ME.SparseTensor(feats, coordinate_manager=st.coordinate_manager, coordinate_map_key=1), 50.0)
        logits = enc(img)
        if gl is None:
            gl = torch.randn(logits.shape, device="cuda", generator=g)
        logits.backward(gl)
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), feats.grad.clone(), {k: p.grad.clone() for k, p in enc.named_parameters()},
                     {k: v.clone() for k, v in enc.state_dict().items()}))
    a, b = runs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_side_streams_leave_every_parameter_bit_identical_at_bench_size():
    """csrc/trunk.hip runs the downsample branch of every layer's first block on the second stream (forward) and on a third
    stream (backward).  Same kernels, same arguments: after 12 optimiser steps on bench scans (bs 2, prefetching, as the
    bench runs) every parameter and running statistic is bit-identical with the side streams on and off (two child
    processes; `scripts/soak_side_streams.py`, 80 steps at bs 4 and bs 2 in profiles/r05_soak_side_streams.txt)."""
    import os
    import subprocess
    import sys
    from helpers import REPO
    p = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "soak_side_streams.py"), "12", "2"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "bit-identical after 12 steps: True" in p.stdout and p.stdout.count("_TrunkFnBackward") == 2
