"""Trunk executor without a GPU: the static program built from the model (lidog_amd/trunk.py) and the C walk in dry
mode (csrc/trunk.hip: table checks, memory plan, which convolutions receive gradients).  No kernel is launched."""
import ctypes

import numpy as np
import pytest


def _tables(prog, levels):
    """per-batch tables with placeholder addresses (dry runs never dereference them)"""
    from lidog_amd import trunk
    A = 4096
    maps = np.zeros((len(prog.map_keys), trunk.TM_COLS), np.int64)
    for i, key in enumerate(prog.map_keys):
        if key[0] == "identity":
            n = levels[int(np.log2(key[1]))]
            maps[i, :14] = [1, n, n, n, 0, 0, 0, 0, 0, 0, A, (n + 127) // 128, 0, A]
        else:
            s_in, s_out, ks, _ = key
            n_in, n_out = levels[int(np.log2(s_in))], levels[int(np.log2(s_out))]
            K = ks ** 3
            P = n_out * 4 if ks == 3 else (n_in if ks == 2 else n_out * 20)
            maps[i, :14] = [K, n_in, n_out, P, A, A, A, A, A, A, A, (P + 127) // 128 + K, A, 0]
    convs = np.zeros((len(prog.convs), trunk.TC_COLS), np.int64)
    for i, (cv, bnm, kind, _, _) in enumerate(prog.convs):
        K = 1 if kind == trunk.KIND_1X1 else cv.kernel_volume
        b = A if cv.bias is not None else 0
        row = [kind, prog.conv_map[i], cv.in_channels, cv.out_channels, K, A, A, A, b, b] + \
              ([A] * 6 if bnm is not None else [0] * 6) + [A, 64, A]
        convs[i, :len(row)] = row
    return maps, convs, np.zeros((len(prog.convs), 2))


@pytest.fixture(scope="module")
def prog():
    import lidog_amd
    from lidog_amd import trunk
    model = lidog_amd.MinkUNet34BEV(1, 7, 3)
    p = trunk.program_of(model)
    assert p is not None
    return p


def test_program_mirrors_the_model(prog):
    from lidog_amd import trunk
    assert len(prog.convs) == 63 and len(prog.bns) == 62 and len(prog.params) == 188
    ops = prog.ops
    assert (ops[:, trunk.TO_COLS - 8] == trunk.OP_CAT).sum() == 4      # 4 skip concatenations
    assert (ops[:, 0] == trunk.OP_CONV).sum() == 1                      # the classifier
    kinds = [k for _, _, k, _, _ in prog.convs]
    assert kinds.count(trunk.KIND_STEM) == 1 and kinds.count(trunk.KIND_DOWN) == 4 and kinds.count(trunk.KIND_UP) == 4
    assert kinds.count(trunk.KIND_1X1) == 7 + 1                         # 7 downsample branches + classifier
    assert kinds.count(trunk.KIND_K3) == 2 * 23
    # every buffer except the input is written exactly once, before it is read
    written = {0}
    for op in ops:
        reads = [op[2]] + ([op[7]] if op[0] == trunk.OP_CAT else []) + ([op[5]] if op[5] >= 0 else [])
        assert all(r in written for r in reads)
        assert op[3] not in written
        written.add(op[3])
    assert written == set(range(len(prog.bufs)))
    # the trace equals what the operator path records (one entry per convolution, forward order)
    assert len(prog.trace) == 63 and prog.trace[0] == ((1, 1, 5, 1), 1, 32) and prog.trace[-1][0] == ("identity", 1)
    # external tensors: block8 features, logits, bottleneck and the three other decoder levels
    assert sorted(prog.ext_shape) == [1, 2, 3, 4, 5, 6]
    assert prog.ext_shape[trunk.EXT_LOGITS] == (0, 7) and prog.ext_shape[trunk.EXT_BOTTLE] == (4, 256)


def _dry(prog, levels, ext_grad, lane=True):
    from lidog_amd import _lib
    L = _lib.load()
    lv = np.array(levels, np.int64)
    maps, convs, conv_f = _tables(prog, levels)
    ext = np.full(7, 4096, np.int64)
    rec = np.zeros(prog.n_rec, np.int64)
    need_f, need_b = np.zeros(2, np.int64), np.zeros(3, np.int64)
    done = np.zeros(len(convs), np.int32)
    common = (convs.ctypes.data, conv_f.ctypes.data, len(convs), maps.ctypes.data, len(maps), prog.ops.ctypes.data,
              len(prog.ops), prog.bufs.ctypes.data, len(prog.bufs), lv.ctypes.data, ext.ctypes.data)
    rc = L.lidog_trunk_forward(*common, None, 0, None, 0, rec.ctypes.data, need_f.ctypes.data, 1, None, None, None)
    assert rc == 0, L.lidog_last_error()
    eg = np.array(ext_grad, np.int64)
    rc = L.lidog_trunk_backward(*common, eg.ctypes.data, None, rec.ctypes.data, None, 0, None, 0, None, 0,
                                need_b.ctypes.data, done.ctypes.data, 1, 0, None, None,
                                ctypes.c_void_p(4096 if lane else 0))
    assert rc == 0, L.lidog_last_error()
    return need_f, need_b, done, common, (rec, lv, maps, convs, conv_f, ext)   # the arrays `common` points into


def test_dry_run_plans_memory_and_gradient_reach(prog):
    small, big = [20000, 12000, 6000, 2500, 900], [352000, 207000, 102000, 41000, 15000]
    both = [0, 4096, 4096, 0, 0, 0, 0]
    f_s, b_s, done, _, _ = _dry(prog, small, both)
    f_b, b_b, _, _, _ = _dry(prog, big, both)
    assert done.all()
    assert (f_b > f_s).all() and (b_b[:2] > b_s[:2]).all()
    assert 4.5e9 < f_b[0] < 6e9 and b_b[0] < 8e9            # ~5 GB of activations, ~6.5 GB of gradients at the bench shape
    # weight gradients in line: their partial slots come out of the main scratch, none on the lane
    _, b_inline, _, _, _ = _dry(prog, small, both, lane=False)
    assert b_inline[2] == 0 and b_inline[1] >= b_s[1]
    # warm-up epochs: no classifier gradient -> the classifier is skipped, everything else still runs
    _, _, done, _, _ = _dry(prog, small, [0, 4096, 0, 0, 0, 0, 0])
    assert done[-1] == 0 and done[:-1].all()
    # segmentation loss only: block8's gradient comes from the classifier alone
    _, _, done, _, _ = _dry(prog, small, [0, 0, 4096, 0, 0, 0, 0])
    assert done.all()
    # nothing reaches the trunk: nothing to do
    _, _, done, _, _ = _dry(prog, small, [0] * 7)
    assert not done.any()


def test_batchnorms_applied_in_their_readers_staging_are_conv1_of_every_block(prog):
    """fusion 4 of the executor (csrc/trunk.hip:in_bn_reader): a BatchNorm + ReLU loses its apply pass exactly when its
    output has one reader and that reader is a 3^3 convolution + BatchNorm with channel counts in multiples of 32 -- in
    MinkUNet34 conv1 -> norm1 -> relu -> conv2 of the 23 BasicBlocks and nothing else (not the stem or the strided
    convolutions, whose outputs a block reads twice, not the block outputs, not the transposed convolutions that feed a
    concatenation)"""
    from lidog_amd import _lib, trunk
    L = _lib.load()
    maps, convs, _ = _tables(prog, [20000, 12000, 6000, 2500, 900])
    ops = prog.ops
    readers = np.full(len(ops), -2, np.int32)
    before = L.lidog_trunk_fusions(7)
    try:
        rc = L.lidog_trunk_in_bn_readers(convs.ctypes.data, len(convs), maps.ctypes.data, len(maps), ops.ctypes.data, len(ops),
                                         prog.bufs.ctypes.data, len(prog.bufs), readers.ctypes.data)
        assert rc == 0, L.lidog_last_error()
        folded = np.nonzero(readers >= 0)[0]
        assert len(folded) == 23
        for o in folded:
            r = readers[o]
            assert r == o + 1 or r == o + 2                       # conv2 follows conv1, a downsample branch may sit between
            assert ops[o, 0] == trunk.OP_CONVBN and ops[o, 4] == 1 and ops[o, 5] < 0          # BatchNorm + ReLU, no residual
            assert ops[r, 0] == trunk.OP_CONVBN and ops[r, 2] == ops[o, 3] and ops[r, 5] >= 0  # reads it; adds the residual
            c1, c2 = convs[ops[o, 1]], convs[ops[r, 1]]
            assert c1[0] == c2[0] == trunk.KIND_K3 and c2[2] == c2[3] == c1[3] and c1[3] % 32 == 0
        L.lidog_trunk_fusions(3)                                    # switched off: nobody
        L.lidog_trunk_in_bn_readers(convs.ctypes.data, len(convs), maps.ctypes.data, len(maps), ops.ctypes.data, len(ops),
                                    prog.bufs.ctypes.data, len(prog.bufs), readers.ctypes.data)
        assert (readers == -1).all()
    finally:
        L.lidog_trunk_fusions(before)


def test_data_parallel_descriptor_is_checked_and_planned(prog):
    """dp argument of the executor (include/lidog_amd.h): SyncBatchNorm without a communicator or callback, and bucket
    tables without a transport, are refused; a dry run with SyncBatchNorm on plans the same regions (the joint
    conv1 + downsample message lives in the scratch region)"""
    from lidog_amd import _lib
    L = _lib.load()
    levels = [20000, 12000, 6000, 2500, 900]
    need_f, need_b, _, common, keep = _dry(prog, levels, [0, 4096, 4096, 0, 0, 0, 0])
    rec, need = np.zeros(prog.n_rec, np.int64), np.zeros(2, np.int64)
    dp = np.zeros(12, np.int64)
    dp[0] = 1
    rc = L.lidog_trunk_forward(*common, None, 0, None, 0, rec.ctypes.data, need.ctypes.data, 1, dp.ctypes.data, None, None)
    assert rc != 0 and b"communicator or a callback" in L.lidog_last_error()
    dp[2] = 4096       # a callback address: never called in a dry run
    rc = L.lidog_trunk_forward(*common, None, 0, None, 0, rec.ctypes.data, need.ctypes.data, 1, dp.ctypes.data, None, None)
    assert rc == 0, L.lidog_last_error()
    assert need[0] == need_f[0] and need_f[1] <= need[1] <= 1.1 * need_f[1]   # conv1 and the downsample share one pass
    dp2 = np.zeros(12, np.int64)
    dp2[6] = 3
    need3, done = np.zeros(3, np.int64), np.zeros(len(prog.convs), np.int32)
    eg = np.array([0, 4096, 4096, 0, 0, 0, 0], np.int64)
    rc = L.lidog_trunk_backward(*common, eg.ctypes.data, None, rec.ctypes.data, None, 0, None, 0, None, 0,
                                need3.ctypes.data, done.ctypes.data, 1, 0, dp2.ctypes.data, None, None)
    assert rc != 0 and b"bucket tables" in L.lidog_last_error()


def test_real_run_refuses_regions_that_are_too_small(prog):
    from lidog_amd import _lib
    L = _lib.load()
    levels = [20000, 12000, 6000, 2500, 900]
    need_f, _, _, common, keep = _dry(prog, levels, [0, 4096, 4096, 0, 0, 0, 0])
    rec = keep[0]
    need = np.zeros(2, np.int64)
    # one byte short of the plan: refused before anything is launched (no GPU is touched on this path)
    rc = L.lidog_trunk_forward(*common, ctypes.c_void_p(4096), int(need_f[0]) - 1, ctypes.c_void_p(4096), int(need_f[1]),
                               rec.ctypes.data, need.ctypes.data, 0, None, None, None)
    assert rc != 0 and b"arena" in L.lidog_last_error()


def test_tables_are_checked(prog):
    from lidog_amd import _lib, trunk
    L = _lib.load()
    levels = [20000, 12000, 6000, 2500, 900]
    lv = np.array(levels, np.int64)
    maps, convs, conv_f = _tables(prog, levels)
    maps[2, 2] += 1          # n_in of the first 3^3 map no longer matches its buffers
    ext = np.full(7, 4096, np.int64)
    rec, need = np.zeros(prog.n_rec, np.int64), np.zeros(2, np.int64)
    rc = L.lidog_trunk_forward(convs.ctypes.data, conv_f.ctypes.data, len(convs), maps.ctypes.data, len(maps),
                               prog.ops.ctypes.data, len(prog.ops), prog.bufs.ctypes.data, len(prog.bufs),
                               lv.ctypes.data, ext.ctypes.data, None, 0, None, 0, rec.ctypes.data, need.ctypes.data, 1,
                               None, None, None)
    assert rc != 0 and b"rows" in L.lidog_last_error()


def test_program_follows_structural_changes_of_the_model():
    """the program holds references to modules and parameters: replacing one (a new classifier head, SyncBatchNorm
    conversion after the first step) must rebuild it, an unchanged model must not"""
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd import trunk
    model = lidog_amd.MinkUNet34BEV(1, 7, 3)
    p0 = trunk.program_of(model)
    assert trunk.program_of(model) is p0
    model.final = ME.MinkowskiConvolution(96, 11, kernel_size=1, bias=True, dimension=3)
    p1 = trunk.program_of(model)
    assert p1 is not p0 and p1.ext_shape[trunk.EXT_LOGITS] == (0, 11) and p1.params[-2] is model.final.kernel
    model = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    p2 = trunk.program_of(model)
    assert p2 is not p1 and all(type(b) is ME.MinkowskiSyncBatchNorm for b in p2.bns)


def test_weight_gradient_work_items_fill_whole_rounds_when_asked(monkeypatch):
    """me._wgrad_chunk: with LIDOG_WGRAD_FIT the items of a launch (one partial item per offset included) fit a whole
    number of rounds of the kernel's resident workgroups, tightly; without it the round-4 rule (pairs / optimum, which
    overshoots by about half an item per offset).  Slots come from the GPU library: stubbed here."""
    import lidog_amd.me as ME
    rng = np.random.default_rng(3)
    # a 3^3 rule book like the bench's stride-1 map: the centre offset owns one pair per voxel, the others 5-40 %
    n = 352_000
    cnt = (n * np.concatenate([rng.uniform(0.05, 0.4, 13), [1.0], rng.uniform(0.05, 0.4, 13)])).astype(np.int64)
    k_off = np.concatenate([[0], np.cumsum(cnt)])
    items = lambda chunk: int(np.sum((cnt + chunk - 1) // chunk))   # noqa: E731
    monkeypatch.setattr(ME, "_wgrad_slots", lambda cin, cout: {96: 512, 128: 768, 256: 768}[cin])
    monkeypatch.setattr(ME, "_WGRAD_FIT", 0)
    old = ME._wgrad_chunk(k_off, 128, 128)
    assert old % 32 == 0 and 2048 - 40 < items(old) < 2048 + 40   # ~2 048 +- partial items: 2.67 rounds of 768 slots
    monkeypatch.setattr(ME, "_WGRAD_FIT", 1)
    for (cin, cout, tiles, slots) in ((96, 96, 1, 512), (128, 128, 1, 768), (256, 256, 4, 768)):
        chunk = ME._wgrad_chunk(k_off, cin, cout)
        wgs = items(chunk) * tiles
        rounds = -(-wgs // slots)
        assert chunk % 32 == 0 and wgs <= rounds * slots
        assert items(chunk - 32) * tiles > rounds * slots or chunk == 128, "not the smallest chunk that fits"
    monkeypatch.setattr(ME, "_WGRAD_FIT", -1)           # default: by stream mode
    monkeypatch.setattr(ME._WgradLane, "enabled", True)
    assert ME._wgrad_chunk(k_off, 128, 128) == old
    monkeypatch.setattr(ME._WgradLane, "enabled", False)
    assert items(ME._wgrad_chunk(k_off, 128, 128)) <= 3 * 768      # 2 048 / 768 = 2.67 -> three whole rounds
