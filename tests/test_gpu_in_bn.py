"""BatchNorm + ReLU applied in the consumer's staging (lidog_sconv_gemm_in_bn / _os_stats_in_bn / _wgrad_in_bn, the
executor's fusion 4) through the C ABI: each form takes the RAW output of the layer before plus that layer's BatchNorm
vectors and must equal `lidog_bn_apply_bits` followed by the plain entry point bit for bit -- gathered GEMM, the
output-stationary kernel with its statistics, and the weight gradient, over every channel-tile shape of the network.
Reference semantics: conv1 -> norm1 -> relu -> conv2 of ME's BasicBlock as called from
utils/models/minkunet_bev.py:312-371."""
import pytest
import torch

from helpers import small_batch

pytestmark = pytest.mark.gpu


def _setup(coords):
    import lidog_amd.me as ME
    st = ME.SparseTensor(coordinates=coords.cuda(), features=torch.ones((coords.shape[0], 1), device="cuda"))
    return ME, st.coordinate_manager


def _bn_inputs(n, C, seed):
    from lidog_amd._lib import call, ptr
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, C, generator=g).cuda()
    mean = (torch.randn(C, generator=g) * 0.3).cuda()
    invstd = (torch.rand(C, generator=g) + 0.5).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = (torch.randn(C, generator=g) * 0.3).cuda()
    y = torch.empty_like(x)
    call("lidog_bn_apply_bits", ptr(x), n, C, 1, ptr(mean), ptr(invstd), ptr(w), ptr(b), None, 1, ptr(y), None)
    assert (y == 0).float().mean().item() > 0.2          # the ReLU bites: the fold must clamp too
    return x, y, (mean, invstd, w, b)


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (64, 64), (96, 96), (128, 128), (256, 256), (128, 96), (32, 64)])
def test_gathered_gemm_and_weight_gradient_with_the_input_batchnorm_folded_in(Cin, Cout):
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    ME, cm = _setup(small_batch((0, 1), n_points=2500))
    m = cm.kernel_map(1, 1, 3)
    n = m.n_out
    x, y, (mean, invstd, w, b) = _bn_inputs(n, Cin, Cin * 7 + Cout)
    g = torch.Generator().manual_seed(Cout)
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    gy = torch.randn(n, Cout, generator=g).cuda()
    # forward product rows
    T1, T2 = torch.empty(m.P, Cout, device="cuda"), torch.empty(m.P, Cout, device="cuda")
    ME._gemm(y, m.pair_in, W, None, m, Cin, Cout, T1, None)
    call("lidog_sconv_gemm_in_bn", ptr(x), ptr(m.pair_in), ptr(W), None, ptr(m.tiles[0]), ptr(m.tiles[1]), ptr(m.tiles[2]),
         m.n_tiles, Cin, Cout, ptr(T2), None, ptr(mean), ptr(invstd), ptr(w), ptr(b), 1, x.shape[0])
    assert torch.equal(T1, T2)
    # weight gradient
    items, n_items, item_off = ME._wgrad_items(m, Cin, Cout)
    slabs = _lib.load().lidog_sconv_wgrad_slabs(Cin, Cout, n_items)
    partial = torch.empty((max(slabs, 1), Cin, Cout), device="cuda")
    g1, g2 = torch.empty_like(W), torch.empty_like(W)
    call("lidog_sconv_wgrad", ptr(y), ptr(m.pair_in), ptr(gy), ptr(m.pair_out), ptr(items), n_items, ptr(item_off), 27, Cin,
         Cout, ptr(partial), ptr(g1))
    partial.fill_(float("nan"))
    call("lidog_sconv_wgrad_in_bn", ptr(x), ptr(m.pair_in), ptr(gy), ptr(m.pair_out), ptr(items), n_items, ptr(item_off), 27,
         Cin, Cout, ptr(partial), ptr(g2), ptr(mean), ptr(invstd), ptr(w), ptr(b), 1)
    assert torch.equal(g1, g2)


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (64, 64), (96, 96), (128, 128)])
def test_output_stationary_kernel_with_the_input_batchnorm_folded_in(Cin, Cout):
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    from test_gpu_sconv_os import _sorted
    L = _lib.load()
    ME, cm = _setup(small_batch((2, 3), n_points=3000))
    m = cm.kernel_map(1, 1, 3)
    n = m.n_out
    perm, wm, order = _sorted(m)
    x, y, (mean, invstd, w, b) = _bn_inputs(n, Cin, Cin * 11 + Cout)
    g = torch.Generator().manual_seed(Cout + 1)
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    outs = []
    for src, fold in ((y, False), (x, True)):
        o = torch.empty(n, Cout, device="cuda")
        su = torch.zeros(2 * Cout + 1, dtype=torch.float64, device="cuda")
        ws = torch.empty(L.lidog_sconv_os_stats_ws(n, Cout), dtype=torch.float64, device="cuda")
        me_, is_ = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
        head = (ptr(src), ptr(m.nbr), n, 27, ptr(perm), ptr(wm), ptr(order), ptr(W), None, Cin, Cout, ptr(o), ptr(su),
                ptr(ws), float(n), 1e-5, 0.1, ptr(me_), ptr(is_), None, None)
        if fold:
            call("lidog_sconv_os_stats_in_bn", *head, ptr(mean), ptr(invstd), ptr(w), ptr(b), 1)
        else:
            call("lidog_sconv_os_stats", *head)
        outs.append((o, su, me_, is_))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


def test_the_folded_forms_refuse_what_they_cannot_do():
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    ME, cm = _setup(small_batch((0,), n_points=500))
    m = cm.kernel_map(1, 1, 3)
    x = torch.zeros(m.n_out, 16, device="cuda")
    W = torch.zeros(27, 16, 32, device="cuda")
    T = torch.empty(m.P, 32, device="cuda")
    v = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError, match="matrix-core kernels only"):     # 16 input channels: no MFMA tile
        call("lidog_sconv_gemm_in_bn", ptr(x), ptr(m.pair_in), ptr(W), None, ptr(m.tiles[0]), ptr(m.tiles[1]),
             ptr(m.tiles[2]), m.n_tiles, 16, 32, ptr(T), None, ptr(v), ptr(v), ptr(v), ptr(v), 1, x.shape[0])
    x = torch.zeros(m.n_out, 32, device="cuda")
    W = torch.zeros(27, 32, 32, device="cuda")
    with pytest.raises(RuntimeError, match="vectors missing"):
        call("lidog_sconv_gemm_in_bn", ptr(x), ptr(m.pair_in), ptr(W), None, ptr(m.tiles[0]), ptr(m.tiles[1]),
             ptr(m.tiles[2]), m.n_tiles, 32, 32, ptr(T), None, None, None, None, None, 1, x.shape[0])
