"""The gathered GEMM with several (tile, column tile) units per workgroup (csrc/sconv_mfma.hip:k_sconv_gemm_mfma_ms,
round 6): the same product rows, bit for bit, as one unit per workgroup -- which tests/test_gpu_ops.py holds against the
oracle's fmaf chains (oracle/me_oracle.c:orc_conv_fwd) -- for every channel-tile shape of MinkUNet34's convolutions
(utils/models/minkunet_bev.py:56-123), with and without a gather index, with the producer's BatchNorm folded into the
staging, and for every way the units can fall onto the workgroups (one workgroup walking all of them, a remainder, more
workgroups than units)."""
import pytest
import torch

from helpers import small_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    import lidog_amd.me as ME
    coords = small_batch((0, 1), n_points=2500)
    st = ME.SparseTensor(coordinates=coords.cuda(), features=torch.ones((coords.shape[0], 1), device="cuda"))
    cm = st.coordinate_manager
    return ME, cm.kernel_map(1, 1, 3)


@pytest.fixture(autouse=True)
def _restore_units():
    from lidog_amd import _lib
    yield
    _lib.load().lidog_sconv_gemm_units(1, 0)


def _product(ME, m, A, gather, W, Cin, Cout, multi, slots):
    from lidog_amd import _lib
    _lib.load().lidog_sconv_gemm_units(multi, slots)
    T = torch.full((m.P, Cout), float("nan"), device="cuda")
    ME._gemm(A, gather, W, None, m, Cin, Cout, T, None)
    return T


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (64, 64), (96, 96), (128, 128), (256, 256), (128, 96), (32, 64), (384, 256),
                                      (192, 128), (96, 32)])
def test_units_per_workgroup_do_not_change_a_bit(scene, Cin, Cout):
    ME, m = scene
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(m.n_in, Cin, generator=g).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    want = _product(ME, m, x, m.pair_in, W, Cin, Cout, 0, 0)
    assert not torch.isnan(want).any()
    n_units = m.n_tiles * (Cout // (128 if Cout % 128 == 0 else 96 if Cout % 96 == 0 else 64 if Cout % 64 == 0 else 32))
    # 1 workgroup for everything; 7 (a remainder); one fewer workgroup than units (a single second unit); the device's own plan
    for slots in (1, 7, max(1, n_units - 1), 0):
        got = _product(ME, m, x, m.pair_in, W, Cin, Cout, 1, slots)
        assert torch.equal(got, want), (Cin, Cout, slots)


def test_without_a_gather_index_rows_are_taken_in_place(scene):
    """the 1x1 convolutions (downsample branches, minkunet_bev.py:414-420) multiply the rows where they stand"""
    ME, _ = scene
    n, Cin, Cout = 5000, 128, 256
    m = ME._IdentityMap(n, "cuda") if hasattr(ME, "_IdentityMap") else None
    if m is None:
        pytest.skip("no identity map type")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, Cin, generator=g).cuda()
    W = (torch.randn(1, Cin, Cout, generator=g) * 0.1).cuda()
    want = _product(ME, m, x, None, W, Cin, Cout, 0, 0)
    for slots in (3, 16):
        assert torch.equal(_product(ME, m, x, None, W, Cin, Cout, 1, slots), want)
    assert torch.allclose(want, x @ W[0], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (96, 96), (128, 128), (256, 256)])
def test_input_batchnorm_fold_with_several_units(scene, Cin, Cout):
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    ME, m = scene
    g = torch.Generator().manual_seed(Cin + 3 * Cout)
    x = torch.randn(m.n_in, Cin, generator=g).cuda()
    mean, invstd = (torch.randn(Cin, generator=g) * 0.3).cuda(), (torch.rand(Cin, generator=g) + 0.5).cuda()
    w, b = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    W = (torch.randn(27, Cin, Cout, generator=g) * 0.1).cuda()
    outs = []
    for multi, slots in ((0, 0), (1, 5), (1, 0)):
        _lib.load().lidog_sconv_gemm_units(multi, slots)
        T = torch.full((m.P, Cout), float("nan"), device="cuda")
        call("lidog_sconv_gemm_in_bn", ptr(x), ptr(m.pair_in), ptr(W), None, ptr(m.tiles[0]), ptr(m.tiles[1]),
             ptr(m.tiles[2]), m.n_tiles, Cin, Cout, ptr(T), None, ptr(mean), ptr(invstd), ptr(w), ptr(b), 1, x.shape[0])
        outs.append(T)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_unknown_or_huge_inputs_keep_one_unit_per_workgroup(scene):
    """a_rows = 0 (not known) must not take the 32-bit row offsets of the multi-unit kernel; the result is the same anyway"""
    from lidog_amd import _lib
    from lidog_amd._lib import call, ptr
    ME, m = scene
    Cin = Cout = 64
    x = torch.randn(m.n_in, Cin, device="cuda")
    W = torch.randn(27, Cin, Cout, device="cuda") * 0.1
    _lib.load().lidog_sconv_gemm_units(1, 2)
    T0, T1 = torch.empty(m.P, Cout, device="cuda"), torch.empty(m.P, Cout, device="cuda")
    for T, rows in ((T0, 0), (T1, x.shape[0])):
        call("lidog_sconv_gemm", ptr(x), ptr(m.pair_in), ptr(W), None, ptr(m.tiles[0]), ptr(m.tiles[1]), ptr(m.tiles[2]),
             m.n_tiles, Cin, Cout, ptr(T), None, rows)
    assert torch.equal(T0, T1)
