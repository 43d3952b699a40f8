"""GPU parity of every HIP operator against the CPU oracle (through the C ABI via lidog_amd.me)."""
import numpy as np
import pytest
import torch

import helpers
from helpers import GOLDEN, small_batch, sha_triples

pytestmark = pytest.mark.gpu


def _rand_coords(seed, n=6000, extent=40, batches=2, dup=False):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(-extent, extent, (n, 3), generator=g, dtype=torch.int32)
    c[:, 2] = torch.randint(-6, 6, (n,), generator=g, dtype=torch.int32)
    b = torch.randint(0, batches, (n, 1), generator=g, dtype=torch.int32)
    c = torch.cat([b, c], dim=1)
    if not dup:
        c = torch.unique(c, dim=0)
        c = c[torch.randperm(c.shape[0], generator=g)]
    return c.contiguous()


def _maps(coords):
    import oracle.me_cpu as OME
    import lidog_amd.me as ME
    so = OME.SparseTensor(coordinates=coords, features=torch.ones(coords.shape[0], 1))
    sg = ME.SparseTensor(coordinates=coords.cuda(), features=torch.ones(coords.shape[0], 1).cuda())
    return so, sg


@pytest.mark.parametrize("seed", [0, 1])
def test_stride_maps_bit_exact(seed):
    coords = _rand_coords(seed)
    so, sg = _maps(coords)
    prev = 1
    for s in (2, 4, 8, 16):
        co = so.coordinate_manager.stride(prev, s)
        cg = sg.coordinate_manager.stride(prev, s).coords
        assert torch.equal(co, cg.cpu()), f"stride {s}: voxel rows differ"
        prev = s


def test_duplicates_first_occurrence():
    import oracle.me_cpu as OME
    import lidog_amd.me as ME
    coords = _rand_coords(3, n=5000, extent=12, dup=True)
    feats = torch.arange(coords.shape[0], dtype=torch.float32).view(-1, 1)
    so = OME.SparseTensor(coordinates=coords, features=feats)
    sg = ME.SparseTensor(coordinates=coords.cuda(), features=feats.cuda())
    assert so.C.shape[0] < coords.shape[0]
    assert torch.equal(so.C, sg.C.cpu())
    assert torch.equal(so.F, sg.F.cpu())


@pytest.mark.parametrize("ks,s_in,s_out", [(3, 1, 1), (5, 1, 1), (2, 1, 2), (3, 2, 2), (2, 2, 4)])
def test_kernel_maps_bit_exact(ks, s_in, s_out):
    coords = _rand_coords(5)
    so, sg = _maps(coords)
    for cm in (so.coordinate_manager, sg.coordinate_manager):
        if s_in > 1:
            cm.stride(1, s_in)
    k_off, pin, pout, nbr = so.coordinate_manager.kernel_map(s_in, s_out, ks)
    km = sg.coordinate_manager.kernel_map(s_in, s_out, ks)
    assert km.k_off_host == k_off.tolist()
    # same pairs in the same order (k-major, ascending out row)
    assert torch.equal(pin, km.pair_in.cpu()) and torch.equal(pout, km.pair_out.cpu())
    assert torch.equal(nbr.t().contiguous(), km.nbr.cpu())
    # position tables invert the pair lists
    P = km.P
    pos_out = km.pos_out.cpu()
    ks_ = np.repeat(np.arange(km.K), np.diff(np.asarray(k_off)))
    assert np.array_equal(pos_out[ks_, pout.numpy()].numpy(), np.arange(P))
    pos_in = km.pos_in.cpu()
    assert np.array_equal(pos_in[ks_, pin.numpy()].numpy(), np.arange(P))
    assert int((pos_out >= 0).sum()) == P and int((pos_in >= 0).sum()) == P


CONV_CASES = [
    # Cin, Cout, ksize, stride, transposed, bias
    (1, 32, 5, 1, False, False),
    (32, 32, 3, 1, False, False),
    (32, 64, 3, 1, False, False),
    (96, 96, 3, 1, False, False),
    (128, 96, 3, 1, False, False),
    (384, 256, 3, 1, False, False),
    (64, 64, 2, 2, False, False),
    (256, 128, 2, 2, True, False),
    (96, 7, 1, 1, False, True),
    (192, 128, 1, 1, False, False),
    (20, 12, 3, 1, False, True),   # generic fallback kernels
    (32, 64, 3, 1, False, True),   # bias gradient wider than 16 columns (BatchNorm statistics reduction)
]


@pytest.fixture(params=[1, 0], ids=["mfma_f32", "vector_fma"])
def sparse_core(request):
    """both arithmetic cores of the sparse GEMMs must give the same (bit-exact) results"""
    from lidog_amd import _lib
    L = _lib.load()
    assert L.lidog_set_sparse_core(request.param) == 0
    yield request.param
    L.lidog_set_sparse_core(1)


@pytest.mark.parametrize("Cin,Cout,ks,stride,transposed,bias", CONV_CASES)
def test_sparse_conv_fwd_bwd(Cin, Cout, ks, stride, transposed, bias, sparse_core):
    import oracle.me_cpu as OME
    import lidog_amd.me as ME
    OME.set_mode("exact")
    coords = _rand_coords(7, n=9000, extent=16)   # ~8.6k voxels: exercises multi-tile and split paths
    so, sg = _maps(coords)
    if transposed:
        so.coordinate_manager.stride(1, 2)
        sg.coordinate_manager.stride(1, 2)
        n_in = so.coordinate_manager.maps[2].shape[0]
        key = 2
    else:
        n_in, key = coords.shape[0], 1
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(n_in, Cin, generator=g)
    cls = "MinkowskiConvolutionTranspose" if transposed else "MinkowskiConvolution"
    co = getattr(OME, cls)(Cin, Cout, kernel_size=ks, stride=stride, bias=bias, dimension=3)
    cg = getattr(ME, cls)(Cin, Cout, kernel_size=ks, stride=stride, bias=bias, dimension=3).cuda()
    cg.load_state_dict(co.state_dict())
    xo = x.clone().requires_grad_(True)
    xg = x.clone().cuda().requires_grad_(True)
    yo = co(OME.SparseTensor(xo, coordinate_manager=so.coordinate_manager, coordinate_map_key=key))
    yg = cg(ME.SparseTensor(xg, coordinate_manager=sg.coordinate_manager, coordinate_map_key=key))
    assert yo.coordinate_map_key == yg.coordinate_map_key
    # forward: identical fmaf chains and identical offset order -> bit-exact
    assert torch.equal(yo.F.detach(), yg.F.detach().cpu()), (yo.F.detach() - yg.F.detach().cpu()).abs().max()
    gy = torch.randn(yo.F.shape, generator=g)
    yo.F.backward(gy)
    yg.F.backward(gy.cuda())
    assert torch.equal(xo.grad, xg.grad.cpu()), (xo.grad - xg.grad.cpu()).abs().max()
    scale = co.kernel.grad.abs().max().item() + 1e-6
    assert (co.kernel.grad - cg.kernel.grad.cpu()).abs().max().item() <= 2e-5 * scale
    if bias:
        torch.testing.assert_close(co.bias.grad, cg.bias.grad.cpu(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,relu,res", [(32, False, False), (96, True, False), (256, True, True), (7, True, True)])
def test_batchnorm_rows(C, relu, res):
    import lidog_amd.me as ME
    g = torch.Generator().manual_seed(C)
    n = 5003
    x = torch.randn(n, C, generator=g) * 2 + 0.5
    r = torch.randn(n, C, generator=g)
    bn_ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5, generator=g)
        bn_ref.bias.normal_(0, 0.2, generator=g)
    mod = ME.MinkowskiBatchNorm(C).cuda()
    mod.bn.load_state_dict(bn_ref.state_dict())
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True)
    y = bn_ref(xr)
    if res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    xg = x.clone().cuda().requires_grad_(True)
    rg = r.clone().cuda().requires_grad_(True)
    yg = ME.batch_norm(xg, mod.bn, 1, relu, rg if res else None)
    torch.testing.assert_close(yg.cpu(), y, rtol=1e-5, atol=2e-6)
    gy = torch.randn(y.shape, generator=g)
    saved = yg.grad_fn.saved_tensors    # freed by backward()
    y.backward(gy)
    yg.backward(gy.cuda())
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(mod.bn.weight.grad.cpu(), bn_ref.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(mod.bn.bias.grad.cpu(), bn_ref.bias.grad, rtol=1e-4, atol=1e-4)
    if res:
        torch.testing.assert_close(rg.grad.cpu(), rr.grad, rtol=0, atol=0)
    torch.testing.assert_close(mod.bn.running_mean.cpu(), bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mod.bn.running_var.cpu(), bn_ref.running_var, rtol=1e-5, atol=1e-6)
    if relu and not res and C % 4 == 0:
        # the backward pass above recomputed the ReLU mask from x; reading it from y instead must give the same bits
        from lidog_amd._lib import call, ptr, load
        xd, dyd = x.cuda(), gy.cuda()
        _, w_s, mean_s, inv_s, y_saved, b_s = saved
        assert y_saved is None and b_s is not None
        yd = yg.detach()
        outs = []
        for use_y in (True, False):
            sums = torch.empty(2 * C + 1, dtype=torch.float64, device="cuda")
            ws = torch.empty(load().lidog_bn_reduce_ws(C, 1), dtype=torch.float64, device="cuda")
            dwb = torch.empty(2, C, device="cuda")
            call("lidog_bn_bwd_reduce", ptr(dyd), ptr(xd), ptr(yd) if use_y else None, n, C, 1, ptr(mean_s), ptr(inv_s),
                 ptr(sums), ptr(ws), float(n), ptr(dwb[0]), ptr(dwb[1]), None if use_y else ptr(w_s), None if use_y else ptr(b_s))
            dx = torch.empty_like(xd)
            call("lidog_bn_bwd_apply", ptr(dyd), ptr(xd), ptr(yd) if use_y else None, n, C, 1, ptr(mean_s), ptr(inv_s),
                 ptr(w_s), ptr(sums), float(n), ptr(dx), None, None, None, None if use_y else ptr(b_s))
            outs.append((sums.clone(), dx, dwb))
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        assert torch.equal(outs[1][1], xg.grad)
    # eval mode uses the running statistics
    bn_ref.eval(), mod.eval()
    torch.testing.assert_close(ME.batch_norm(x.cuda(), mod.bn, 1, False, None).cpu(), bn_ref(x), rtol=1e-5, atol=2e-6)


def test_relu_add_cat():
    import lidog_amd.me as ME
    coords = _rand_coords(11, n=1000)
    x = torch.randn(coords.shape[0], 24)
    a = ME.SparseTensor(coordinates=coords.cuda(), features=x.cuda().requires_grad_(True))
    b = ME.SparseTensor(x.cuda() * 2, coordinate_manager=a.coordinate_manager, coordinate_map_key=1)
    c = ME.cat(a, b)
    assert c.F.shape[1] == 48 and torch.equal(c.F.cpu(), torch.cat([x, 2 * x], dim=1))
    # backward of cat: two contiguous gradients (HIP split kernel), equal to torch's
    xa, xb = x.cuda().requires_grad_(True), (3 * x).cuda().requires_grad_(True)
    ca = ME.cat(ME.SparseTensor(xa, coordinate_manager=a.coordinate_manager, coordinate_map_key=1),
                ME.SparseTensor(xb, coordinate_manager=a.coordinate_manager, coordinate_map_key=1))
    w = torch.randn(coords.shape[0], 48, device="cuda")
    (ca.F * w).sum().backward()
    assert torch.equal(xa.grad, w[:, :24]) and torch.equal(xb.grad, w[:, 24:])
    r = ME.MinkowskiReLU(inplace=True)(a)
    assert torch.equal(r.F.cpu(), torch.relu(x))
    s = a + b
    assert torch.equal(s.F.cpu(), x + 2 * x)
    s.F.sum().backward()
    with pytest.raises(ValueError):
        other = ME.SparseTensor(coordinates=coords.cuda(), features=x.cuda())
        ME.cat(a, other)
    with pytest.raises(ValueError):
        ME.SparseTensor(coordinates=coords.long().cuda(), features=x.cuda())
    with pytest.raises(RuntimeError):
        ME.SparseTensor(coordinates=coords, features=x)  # CPU tensors: no CPU path


def test_coordinate_range_checked():
    import lidog_amd.me as ME
    coords = torch.tensor([[0, 0, 0, 0], [0, 70000, 0, 0]], dtype=torch.int32).cuda()
    with pytest.raises(ValueError):
        ME.SparseTensor(coordinates=coords, features=torch.ones(2, 1).cuda())


@pytest.mark.parametrize("C,mask", [(96, "from_x"), (96, "from_y"), (32, "none"), (256, "from_y"), (96, "bits"),
                                    (36, "bits")])
def test_reduction_with_batchnorm_backward_statistics_in_its_epilogue(C, mask):
    """lidog_sconv_reduce_rows_bwdstats == lidog_sconv_reduce_rows followed by lidog_bn_bwd_reduce on its output: the
    gradient rows, the fp64 sums, the row count behind them and the parameter gradients, bit for bit"""
    import lidog_amd.me as ME
    from lidog_amd._lib import call, load, ptr
    coords = _rand_coords(9, n=30000, extent=24)
    _, sg = _maps(coords)
    m = sg.coordinate_manager.kernel_map(1, 1, 3)
    n = m.n_in
    g = torch.Generator(device="cuda").manual_seed(C)
    T = torch.randn(m.P, C, device="cuda", generator=g)
    addend = torch.randn(n, C, device="cuda", generator=g)
    pre = torch.randn(n, C, device="cuda", generator=g) * 2 + 0.5
    mean = torch.randn(C, device="cuda", generator=g) * 0.1 + 0.5
    invstd = torch.rand(C, device="cuda", generator=g) + 0.5
    w = torch.rand(C, device="cuda", generator=g) + 0.5
    b = torch.randn(C, device="cuda", generator=g) * 0.3
    y = torch.relu((pre - mean) * invstd * w + b + torch.randn(n, C, device="cuda", generator=g))
    rp, rl = m.rows("in")
    L = load()
    ws = torch.empty(L.lidog_bn_reduce_ws(C, 1), dtype=torch.float64, device="cuda")
    ry, rw, rb = (y, None, None) if mask == "from_y" else (None, w, b) if mask == "from_x" else (None, None, None)
    bits = None
    if mask == "bits":
        # the mask as the forward pass leaves it (lidog_bn_apply_bits with a residual): same y, 1/32 of the bytes
        res = torch.randn(n, C, device="cuda", generator=g)
        bits = torch.full((L.lidog_relu_bits_words(n, C),), -1, dtype=torch.int32, device="cuda")
        y2 = torch.empty(n, C, device="cuda")
        call("lidog_bn_apply_bits", ptr(pre), n, C, 1, ptr(mean), ptr(invstd), ptr(w), ptr(b), ptr(res), 1, ptr(y2), ptr(bits))
        y3 = torch.empty(n, C, device="cuda")
        call("lidog_bn_apply", ptr(pre), n, C, 1, ptr(mean), ptr(invstd), ptr(w), ptr(b), ptr(res), 1, ptr(y3))
        assert torch.equal(y2, y3)
        y = y2
        # every bit is the comparison y > 0 of its element
        q = torch.arange(n * C // 4, device="cuda")
        nib = (bits.long()[q >> 3] >> (4 * (q & 7))) & 15
        want = ((y.view(-1, 4) > 0).long() * torch.tensor([1, 2, 4, 8], device="cuda")).sum(1)
        assert torch.equal(nib, want)
    for add in (addend, None):
        out_a, out_b = torch.empty(n, C, device="cuda"), torch.empty(n, C, device="cuda")
        sums_a = torch.full((2 * C + 1,), -1.0, dtype=torch.float64, device="cuda")
        sums_b = sums_a.clone()
        dw_a, db_a, dw_b, db_b = (torch.empty(C, device="cuda") for _ in range(4))
        call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, C, None, ptr(add), ptr(out_a))
        # reference: the two-pass path reading the saved output y (bits mode: y is what the bits were made from)
        call("lidog_bn_bwd_reduce", ptr(out_a), ptr(pre), ptr(y if mask == "bits" else ry), n, C, 1, ptr(mean), ptr(invstd),
             ptr(sums_a), ptr(ws), float(n), ptr(dw_a), ptr(db_a), ptr(rw), ptr(rb))
        call("lidog_sconv_reduce_rows_bwdstats", ptr(T), ptr(rp), ptr(rl), n, C, ptr(add), ptr(out_b), ptr(pre), ptr(ry),
             ptr(bits), ptr(mean), ptr(invstd), ptr(rw), ptr(rb), ptr(sums_b), ptr(ws), float(n), ptr(dw_b), ptr(db_b))
        assert torch.equal(out_a, out_b)
        assert torch.equal(sums_a, sums_b) and float(sums_b[2 * C]) == n
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
        if mask == "bits":   # the two-pass kernels with the bit mask: same sums, same dx / dres as with y
            sums_c = torch.empty_like(sums_a)
            call("lidog_bn_bwd_reduce_bits", ptr(out_a), ptr(pre), None, ptr(bits), n, C, 1, ptr(mean), ptr(invstd),
                 ptr(sums_c), ptr(ws), float(n), ptr(dw_b), ptr(db_b), None, None)
            assert torch.equal(sums_c, sums_a)
            dx_a, dx_b, dr_a, dr_b = (torch.empty(n, C, device="cuda") for _ in range(4))
            call("lidog_bn_bwd_apply", ptr(out_a), ptr(pre), ptr(y), n, C, 1, ptr(mean), ptr(invstd), ptr(w), ptr(sums_a),
                 float(n), ptr(dx_a), ptr(dr_a), None, None, None)
            call("lidog_bn_bwd_apply_bits", ptr(out_a), ptr(pre), None, ptr(bits), n, C, 1, ptr(mean), ptr(invstd), ptr(w),
                 ptr(sums_a), float(n), ptr(dx_b), ptr(dr_b), None, None, None)
            assert torch.equal(dx_a, dx_b) and torch.equal(dr_a, dr_b)
        # and the sums are what they should be (float64 reference)
        gm = out_a.double() * ((y > 0) if mask in ("from_y", "bits") else ((pre - mean) * invstd * w + b > 0) if mask == "from_x"
                               else torch.ones_like(y, dtype=torch.bool))
        xh = ((pre - mean) * invstd).double()
        ref = torch.cat([gm.sum(0), (gm * xh).sum(0)])
        torch.testing.assert_close(sums_b[:2 * C], ref, rtol=1e-9, atol=1e-7)


def test_bottleneck_block_matches_the_oracle():
    """ME.modules.resnet_block.Bottleneck (imported by minkunet_bev.py:4; 1x1 -> 3^3 -> 1x1 x 4 + residual through a
    1x1 downsample): forward and every parameter gradient against the oracle's block, training-mode BatchNorm"""
    import oracle.me_cpu as OME
    import lidog_amd.me as ME
    OME.set_mode("exact")
    coords = _rand_coords(17, n=5000, extent=14)
    so, sg = _maps(coords)
    n = coords.shape[0]
    torch.manual_seed(4)
    down_o = torch.nn.Sequential(OME.MinkowskiConvolution(32, 64, kernel_size=1, dimension=3), OME.MinkowskiBatchNorm(64))
    blk_o = OME.modules.resnet_block.Bottleneck(32, 16, downsample=down_o, dimension=3).train()
    down_g = torch.nn.Sequential(ME.MinkowskiConvolution(32, 64, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(64))
    blk_g = ME.modules.resnet_block.Bottleneck(32, 16, downsample=down_g, dimension=3)
    assert list(blk_g.state_dict().keys()) == list(blk_o.state_dict().keys())
    blk_g.load_state_dict(blk_o.state_dict())
    blk_g.cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 32, generator=g)
    gy = torch.randn(n, 64, generator=g)
    xo, xg = x.clone().requires_grad_(True), x.clone().cuda().requires_grad_(True)
    yo = blk_o(OME.SparseTensor(xo, coordinate_manager=so.coordinate_manager, coordinate_map_key=1))
    yg = blk_g(ME.SparseTensor(xg, coordinate_manager=sg.coordinate_manager, coordinate_map_key=1))
    assert (yo.F.detach() - yg.F.detach().cpu()).abs().max().item() <= 2e-5
    yo.F.backward(gy)
    yg.F.backward(gy.cuda())
    assert (xo.grad - xg.grad.cpu()).abs().max().item() <= 2e-5 * xo.grad.abs().max().item() + 1e-7
    for (k, po), (_, pg) in zip(blk_o.named_parameters(), blk_g.named_parameters()):
        a, b = pg.grad.cpu().double().flatten(), po.grad.double().flatten()
        assert float((a - b).norm() / (b.norm() + 1e-30)) <= 1e-4, k


_TAIL_CODE = r"""
import sys, torch
sys.path.insert(0, %r)
from lidog_amd import _lib
from lidog_amd._lib import call, ptr
L = _lib.load()
out = []
for n, C in ((400000, 96), (1000003, 32), (70001, 256)):
    g = torch.Generator().manual_seed(n + C)
    x = (torch.randn(n, C, generator=g) * 3 + 1).cuda()
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.lidog_bn_reduce_ws(C, 1), dtype=torch.float64, device="cuda")
    mean, invstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    for rep in range(3):
        call("lidog_bn_stats", ptr(x), n, C, 1, ptr(sums), ptr(ws), float(n), 1e-5, 0.1, ptr(mean), ptr(invstd), None, None)
    out.append(torch.cat([sums, mean.double(), invstd.double()]).cpu())
torch.save(out, sys.argv[1])
"""


def test_in_kernel_statistics_tail_equals_the_separate_finish_kernel(tmp_path):
    """the last-arriver tail of the statistics reductions (csrc/stats_tail.h: write-through stores, one ticket
    add per workgroup) against bn.hip:k_sums_finish in a launch of its own (LIDOG_STATS_TAIL=0, read once per process:
    two child processes) on grids that span every XCD -- sums, mean and invstd to 1e-12, three launches in a row on
    one stream (the ticket words must come back to zero)."""
    import os
    import subprocess
    import sys
    from helpers import REPO
    res = {}
    for tail in ("1", "0"):
        f = str(tmp_path / f"tail{tail}.pt")
        env = dict(os.environ, LIDOG_STATS_TAIL=tail)
        p = subprocess.run([sys.executable, "-c", _TAIL_CODE % REPO, f], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tail] = torch.load(f)
    # the two finishes add the same per-workgroup partial rows in different groupings (32-row groups vs 64 strided
    # chains), so the float64 sums agree to rounding, not bit for bit; a partial row read STALE by the last arriver (the
    # failure the hand-off protocol must exclude) would be off by ~1/workgroups, eleven orders of magnitude more
    for a, b in zip(res["1"], res["0"]):
        assert ((a - b).abs() <= 1e-12 * b.abs() + 1e-12).all(), (a - b).abs().max()


def test_first_statistics_launch_on_a_new_stream_while_the_default_stream_is_busy():
    """Round 5 bug: the ticket words of a stream's statistics tail were zeroed with hipMemset, which is asynchronous to
    the host and ordered on the NULL stream only; with a backlog on the NULL stream the first reduction on a NEW
    non-blocking stream started on garbage tickets, no workgroup took itself for the last one, and sums / mean / invstd
    were never written (first downsample branch of a fresh process on the side stream: zero BatchNorm gradients, 1 in 15).
    Eight fresh streams, each with its first launch behind a backlog of matrix products on the default stream; checked
    against the old hipMemset build: fails there (sums stay NaN), passes with hipMemsetAsync on the tickets' own stream."""
    import ctypes
    from lidog_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    n, C = 200_000, 96
    x = torch.randn(n, C, generator=g).cuda()
    ref = torch.cat([x.double().sum(0), (x.double() ** 2).sum(0)])
    a = torch.randn(4096, 4096, device="cuda")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    torch.cuda.synchronize()
    # fresh device memory comes zero-filled from the driver, which hides the bug: make the runtime's allocator hand out
    # RECYCLED fragments instead (filled with ones, freed, then taken again by the ticket allocations below)
    hip = ctypes.CDLL("libamdhip64.so")
    dirty = []
    for _ in range(64):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(768)) == 0
        assert hip.hipMemset(p, 0xFF, ctypes.c_size_t(768)) == 0
        dirty.append(p)
    assert hip.hipDeviceSynchronize() == 0
    for p in dirty[1:-1]:      # the first and the last stay allocated: an empty block would go back to the driver
        assert hip.hipFree(p) == 0
    streams = []
    for _ in range(8):
        st = torch.cuda.Stream()
        streams.append(st)
        sums = torch.full((2 * C + 1,), float("nan"), dtype=torch.float64, device="cuda")
        mean = torch.full((C,), float("nan"), device="cuda")
        invstd = torch.full((C,), float("nan"), device="cuda")
        ws = torch.empty(int(lib.lidog_bn_reduce_ws(C, 1)), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(20):      # ~20 ms of work queued on the default (NULL) stream
            a @ a
        _lib.call_on(st.cuda_stream, "lidog_bn_stats", ptr(x), n, C, 1, ptr(sums), ptr(ws), float(n), 1e-5, 0.1,
                     ptr(mean), ptr(invstd), None, None)
        st.synchronize()
        assert torch.isfinite(sums).all() and torch.isfinite(mean).all() and torch.isfinite(invstd).all(), \
            "the reduction's last workgroup never finished the sums"
        assert ((sums[:2 * C] - ref).abs() <= 1e-9 * ref.abs() + 1e-9).all()
        assert float(sums[2 * C]) == n
        torch.cuda.synchronize()
    for p in (dirty[0], dirty[-1]):
        assert hip.hipFree(p) == 0
