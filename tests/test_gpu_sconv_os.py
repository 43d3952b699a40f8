"""Output-stationary 3^3 convolution (csrc/sconv_os.hip: rows sorted by neighbour mask, no product rows) through the C
ABI: bit-identical to the two-pass path (gathered GEMM -> product rows -> per-row reduction) and to the CPU oracle's
fmaf chains, forward and data gradient, with bias / addend, on small scenes and on a bench-size map; the statistics
forms against the two-pass forms; the sorted-row tables themselves against a numpy restatement.
Reference semantics: MinkowskiConvolution(kernel_size=3, stride=1), utils/models/minkunet_bev.py:425-439."""
import numpy as np
import pytest
import torch

from helpers import small_batch

pytestmark = pytest.mark.gpu


def _setup(coords):
    import lidog_amd.me as ME
    st = ME.SparseTensor(coordinates=coords.cuda(), features=torch.ones((coords.shape[0], 1), device="cuda"))
    return ME, st.coordinate_manager


def _sorted(m):
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    n = m.n_out
    pad = (n + 127) // 128 * 128
    perm = torch.empty(pad, dtype=torch.int32, device="cuda")
    wm = torch.empty(pad // 32, dtype=torch.int32, device="cuda")
    order = torch.empty(pad // 128, dtype=torch.int32, device="cuda")
    ws = torch.empty(_lib.load().lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device="cuda")
    call("lidog_kernel_map_sorted", ptr(m.nbr), n, m.K, ptr(m.k_off), ptr(perm), ptr(wm), ptr(order), ptr(ws), ws.numel())
    return perm, wm, order


def test_sorted_rows_tables_match_their_definition():
    ME, cm = _setup(small_batch((3, 4), n_points=2500))
    m = cm.kernel_map(1, 1, 3)
    perm, wm, order = _sorted(m)
    n = m.n_out
    nbr = m.nbr.cpu().numpy()
    mask = np.zeros(n, np.int64)
    for k in range(27):
        mask |= (nbr[k] >= 0).astype(np.int64) << k
    counts = (nbr >= 0).sum(1)
    # bit position of offset k in the key: number of offsets that occur more often (ties: lower k first)
    pos = [sum(1 for j in range(27) if counts[j] > counts[k] or (counts[j] == counts[k] and j < k)) for k in range(27)]
    key = np.zeros(n, np.int64)
    for k in range(27):
        key |= ((mask >> k) & 1) << pos[k]
    want = np.argsort(key, kind="stable")
    p = perm.cpu().numpy()
    assert np.array_equal(p[:n], want) and (p[n:] == -1).all()
    pad = p.shape[0]
    mm = np.concatenate([mask[want], np.zeros(pad - n, np.int64)]).reshape(-1, 32)
    assert np.array_equal(wm.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, np.bitwise_or.reduce(mm, axis=1))
    wo = np.bitwise_or.reduce(mm, axis=1)                      # per 32 sorted rows
    w = np.array([sum(bin(int(v)).count("1") for v in wo[4 * t:4 * t + 4]) for t in range(pad // 128)])
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(pad // 128))
    assert all(w[o[i]] >= w[o[i + 1]] for i in range(len(o) - 1))      # heaviest tiles first


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (96, 96), (128, 96), (64, 128)])
def test_forward_and_data_gradient_equal_the_two_pass_path_and_the_oracle(Cin, Cout):
    import oracle.me_cpu as OME
    from lidog_amd._lib import call, ptr
    coords = small_batch((0, 1), n_points=3000)
    ME, cm = _setup(coords)
    m = cm.kernel_map(1, 1, 3)
    n = m.n_out
    perm, wm, order = _sorted(m)
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(n, Cin, generator=g).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    gy = torch.randn(n, Cout, generator=g).cuda()
    ad = torch.randn(n, Cin, generator=g).cuda()
    Wt = W.transpose(1, 2).contiguous()
    # two-pass path
    T = torch.empty(m.P, Cout, device="cuda")
    out2 = torch.empty(n, Cout, device="cuda")
    ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    rp, rl = m.rows("out")
    call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, Cout, ptr(bias), None, ptr(out2))
    out = torch.empty(n, Cout, device="cuda")
    call("lidog_sconv_os", ptr(x), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(W), 0, ptr(bias), None, Cin,
         Cout, ptr(out))
    assert torch.equal(out, out2)
    T2 = torch.empty(m.P, Cin, device="cuda")
    gx2 = torch.empty(n, Cin, device="cuda")
    ME._gemm(gy, m.pair_out, Wt, None, m, Cout, Cin, T2, None)
    rpi, rli = m.rows("in")
    call("lidog_sconv_reduce_rows", ptr(T2), ptr(rpi), ptr(rli), n, Cin, None, ptr(ad), ptr(gx2))
    gx = torch.empty(n, Cin, device="cuda")
    call("lidog_sconv_os", ptr(gy), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(Wt), 1, None, ptr(ad), Cout,
         Cin, ptr(gx))
    assert torch.equal(gx, gx2)
    # the CPU oracle's exact mode (fmaf chains over ascending input channel, products summed over ascending offset)
    OME.set_mode("exact")
    st = OME.SparseTensor(coordinates=coords, features=x.cpu())
    conv = OME.MinkowskiConvolution(Cin, Cout, kernel_size=3, stride=1, bias=True, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(W.cpu())
        conv.bias.copy_(bias.cpu().view(1, -1))
    xin = st.F.clone().requires_grad_(True)
    ref = conv(OME.SparseTensor(features=xin, coordinate_manager=st.coordinate_manager, coordinate_map_key=st.coordinate_map_key))
    assert torch.equal(ref.F.detach(), out.cpu())
    ref.F.backward(gy.cpu())
    assert torch.equal(xin.grad + ad.cpu(), gx.cpu())


def test_statistics_form_and_bench_size_map():
    from lidog_amd import _lib, synth
    from lidog_amd._lib import call, ptr
    L = _lib.load()
    b = synth.make_batch(range(2), "kitti120k", "cuda")
    ME, cm = _setup(b["coords_int"].cpu())
    m = cm.kernel_map(1, 1, 3)
    # which maps the product path hands to this kernel: sparse (<= 6 pairs per row) and large (>= 1500 tiles) ones
    assert (m.sorted() is not None) == (m.P <= ME._SCONV_OS_DENSITY * m.n_out and m.n_out >= 128 * ME._SCONV_OS_MIN_TILES)
    perm, wm, order = _sorted(m)
    n, Cin, Cout = m.n_out, 96, 96
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, Cin, generator=g).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    T = torch.empty(m.P, Cout, device="cuda")
    o1, o2 = torch.empty(n, Cout, device="cuda"), torch.empty(n, Cout, device="cuda")
    su1 = torch.empty(2 * Cout + 1, dtype=torch.float64, device="cuda")
    su2 = torch.empty_like(su1)
    ws1 = torch.empty(L.lidog_sconv_reduce_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
    ws2 = torch.empty(L.lidog_sconv_os_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
    me1, is1, me2, is2 = (torch.empty(Cout, device="cuda") for _ in range(4))
    rp, rl = m.rows("out")
    ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    call("lidog_sconv_reduce_rows_stats", ptr(T), ptr(rp), ptr(rl), n, Cout, None, ptr(o1), ptr(su1), ptr(ws1), float(n),
         1e-5, 0.1, ptr(me1), ptr(is1), None, None)
    for rep in range(3):       # the ticket words must be left zeroed: three launches on one stream, same result
        su2.zero_()
        call("lidog_sconv_os_stats", ptr(x), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(W), None, Cin, Cout,
             ptr(o2), ptr(su2), ptr(ws2), float(n), 1e-5, 0.1, ptr(me2), ptr(is2), None, None)
        assert torch.equal(o1, o2)
        ref = torch.cat([o2.double().sum(0), (o2.double() ** 2).sum(0)])
        assert ((su2[:-1] - ref).abs() <= 1e-10 * ref.abs() + 1e-9).all() and su2[-1].item() == n
        assert torch.allclose(me1, me2, rtol=1e-6, atol=1e-7) and torch.allclose(is1, is2, rtol=1e-6)
        if rep:
            assert torch.equal(su2, keep)      # run-to-run identical sums
        keep = su2.clone()


def test_evaluation_mode_batchnorm_epilogue_equals_the_two_pass_form():
    """validation path: convolution + evaluation-mode BatchNorm + residual + ReLU in one launch, both forms, same bits"""
    from lidog_amd._lib import call, ptr
    coords = small_batch((5, 6), n_points=3000)
    ME, cm = _setup(coords)
    m = cm.kernel_map(1, 1, 3)
    n, Cin, Cout = m.n_out, 64, 96
    perm, wm, order = _sorted(m)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, Cin, generator=g).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    mean, var = torch.randn(Cout, generator=g).cuda(), (torch.rand(Cout, generator=g) + 0.5).cuda()
    w, b = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    res = torch.randn(n, Cout, generator=g).cuda()
    invstd = torch.empty(Cout, device="cuda")
    call("lidog_bn_eval_invstd", ptr(var), 1e-5, Cout, ptr(invstd))
    T = torch.empty(m.P, Cout, device="cuda")
    ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    rp, rl = m.rows("out")
    for residual, relu in ((res, 1), (None, 1), (None, 0)):
        o1, o2 = torch.empty(n, Cout, device="cuda"), torch.empty(n, Cout, device="cuda")
        call("lidog_sconv_reduce_rows_bn", ptr(T), ptr(rp), ptr(rl), n, Cout, None, ptr(mean), ptr(invstd), ptr(w), ptr(b),
             ptr(residual), relu, ptr(o1))
        call("lidog_sconv_os_bn", ptr(x), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(W), None, Cin, Cout,
             ptr(mean), ptr(invstd), ptr(w), ptr(b), ptr(residual), relu, ptr(o2))
        assert torch.equal(o1, o2)


def test_statistics_form_past_4096_tiles_takes_the_separate_finish():
    """one partial row per 128-row tile, and the in-kernel tail covers 128 x 32 of them; six bench scans
    (533 k rows = 4 167 tiles; `bench.py --batch 6`, or two 0.02 m stress scans) used to raise "too many tiles" inside the
    training step.  Past that size the rows are added by k_sums_finish in a launch of its own: same convolution bits as
    the two-pass path, sums equal to the float64 column sums of the output, run-to-run identical."""
    from lidog_amd import _lib, synth
    from lidog_amd._lib import call, ptr
    L = _lib.load()
    b = synth.make_batch(range(6), "kitti120k", "cuda")
    ME, cm = _setup(b["coords_int"].cpu())
    m = cm.kernel_map(1, 1, 3)
    n, Cin, Cout = m.n_out, 32, 32
    assert (n + 127) // 128 > 4096
    perm, wm, order = _sorted(m)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, Cin, generator=g).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    T = torch.empty(m.P, Cout, device="cuda")
    o1, o2 = torch.empty(n, Cout, device="cuda"), torch.empty(n, Cout, device="cuda")
    rp, rl = m.rows("out")
    ME._gemm(x, m.pair_in, W, None, m, Cin, Cout, T, None)
    call("lidog_sconv_reduce_rows", ptr(T), ptr(rp), ptr(rl), n, Cout, None, None, ptr(o1))
    su = torch.empty(2 * Cout + 1, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.lidog_sconv_os_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
    me, inv = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    keep = None
    for rep in range(2):
        su.zero_()
        call("lidog_sconv_os_stats", ptr(x), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(W), None, Cin, Cout,
             ptr(o2), ptr(su), ptr(ws), float(n), 1e-5, 0.1, ptr(me), ptr(inv), None, None)
        assert torch.equal(o1, o2)
        ref = torch.cat([o2.double().sum(0), (o2.double() ** 2).sum(0)])
        assert ((su[:-1] - ref).abs() <= 1e-10 * ref.abs() + 1e-9).all() and su[-1].item() == n
        assert torch.allclose(me, o2.double().mean(0).float(), rtol=1e-5, atol=1e-6)
        if keep is not None:
            assert torch.equal(su, keep)
        keep = su.clone()
    # the small launch right behind it still finds its stream's ticket words zeroed (the in-kernel tail)
    test_statistics_form_and_bench_size_map()
