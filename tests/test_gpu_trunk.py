"""Trunk executor (csrc/trunk.hip, lidog_amd/trunk.py) against the operator path (lidog_amd/me.py): the same entry
points in the same order, so EVERYTHING must be bit-identical -- outputs, losses, every parameter gradient, running
statistics, and the parameters after optimiser steps -- with the weight gradients on the second stream and in line,
with and without the optimiser's flat buffers, during warm-up (no classifier gradient) and when the model is called
twice before one backward pass."""
import pytest
import torch

from helpers import seeded_state_dict, small_batch

pytestmark = pytest.mark.gpu


def _batch(seeds=(61, 62), n_points=2500, bev=17):
    coords = small_batch(seeds, n_points=n_points).cuda()
    g = torch.Generator().manual_seed(seeds[0])
    n = coords.shape[0]
    return {"coords_int": coords, "source_coordinates0": coords.float(),
            "source_features0": torch.ones((n, 1), device="cuda"),
            "source_sem_labels0": torch.randint(-1, 7, (n,), generator=g).cuda(),
            "source_bev_labels0": {"block8": torch.randint(-1, 7, (len(seeds), bev, bev), generator=g).cuda()}}


def _model(seed=5):
    import lidog_amd
    m = lidog_amd.MinkUNet34BEV(1, 7, 3, mapping_bound_2d=5.0).cuda()
    m.load_state_dict(seeded_state_dict(m, seed))
    return m.train()


def _grads(model):
    return {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}


def _buffers(model):
    return {n: b.clone() for n, b in model.state_dict().items()}


def _assert_same(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, f"{what} {k}: one side has no gradient"
        else:
            assert torch.equal(a[k], b[k]), f"{what} {k}: max |diff| {(a[k] - b[k]).abs().max().item():.3e}"


@pytest.mark.parametrize("fusions,os_mode", [(7, 1), (3, 1), (1, 1), (4, 1), (0, 1), (7, 2), (0, 2)],
                         ids=["fused", "no_input_bn_fold", "bwdstats_only", "input_bn_fold_only", "plain_sequence",
                              "fused_output_stationary", "plain_output_stationary"])
@pytest.mark.parametrize("overlap", [True, False])
def test_executor_step_is_bit_identical_to_the_operator_path(overlap, fusions, os_mode, monkeypatch):
    """3 optimiser steps of the LiDOG step (Adam on flat buffers), executor on vs off; with the executor's fusions
    (BatchNorm-backward statistics in the epilogue of the producing data-gradient reduction, ReLU masks of the residual
    layers as bits, the BatchNorm + ReLU between the two convolutions of a block applied in the second one's staging) and
    without them; os_mode 2: every symmetric 3^3 map takes the output-stationary convolution
    (csrc/sconv_os.hip; by default only maps of >= 1500 tiles do, which these scenes are not)"""
    from lidog_amd import me as ME, trunk
    from lidog_amd.trainer import LiDOGStep
    from lidog_amd.optim import make_optimizer
    monkeypatch.setattr(ME, "_SCONV_OS", os_mode)
    ME.set_backward_overlap(overlap)
    before = trunk.set_fusions(fusions)
    try:
        runs = {}
        for on in (False, True):
            trunk.set_enabled(on)
            model = _model()
            step = LiDOGStep(model, make_optimizer("Adam", model, 1e-2, weight_decay=1e-4))
            losses, grads = [], None
            for it in range(3):
                out = step.training_step(_batch((61 + it, 71 + it)))
                losses.append([float(out[k]) for k in ("loss", "sem_loss", "bev_loss")])
                if it == 0:
                    grads = _grads(model)
            torch.cuda.synchronize()
            runs[on] = (losses, grads, _buffers(model))
        assert runs[True][0] == runs[False][0], f"losses differ: {runs[True][0]} vs {runs[False][0]}"
        _assert_same(runs[True][1], runs[False][1], "gradient")
        _assert_same(runs[True][2], runs[False][2], "state after 3 steps")
    finally:
        trunk.set_enabled(True)
        trunk.set_fusions(before)
        ME.set_backward_overlap(True)


def _forward_backward(model, batch, with_seg=True, with_bev=True):
    import lidog_amd.me as ME
    from lidog_amd.losses import DICELoss, SoftDICELoss
    st = ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"])
    sem, bev = model(st, is_train=True)
    loss = 0.0
    if with_seg:
        loss = loss + SoftDICELoss(ignore_label=-1)(sem.F, batch["source_sem_labels0"].long())
    if with_bev:
        loss = loss + DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7), batch["source_bev_labels0"]["block8"].view(-1))
    loss.backward()
    return sem.F.detach().clone(), float(loss.detach())


@pytest.mark.parametrize("with_seg,with_bev", [(True, True), (False, True), (True, False)])
def test_executor_without_flat_buffers_and_with_missing_loss_terms(with_seg, with_bev):
    """plain parameters (no optimiser yet: gradients come back as fresh tensors); without the segmentation loss the
    classifier gets NO gradient (warm-up epochs, trainer_lighting_2d.py:193-201), without the BEV loss block8's only
    gradient is the classifier's"""
    from lidog_amd import trunk
    batch = _batch()
    res = {}
    try:
        for on in (False, True):
            trunk.set_enabled(on)
            model = _model()
            logits, loss = _forward_backward(model, batch, with_seg, with_bev)
            torch.cuda.synchronize()
            res[on] = (logits, loss, _grads(model), _buffers(model))
    finally:
        trunk.set_enabled(True)
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][1] == res[False][1]
    _assert_same(res[True][2], res[False][2], "gradient")
    _assert_same(res[True][3], res[False][3], "buffer")
    if not with_seg:
        assert res[True][2]["final.kernel"] is None and res[True][2]["final.bias"] is None


def test_executor_was_taken_and_falls_back_where_it_must():
    """the executor really runs in a training step (one autograd node for the trunk) and steps aside for evaluation,
    no_grad and a frozen parameter; a second call before backward gets its own activation arena"""
    import lidog_amd.me as ME
    from lidog_amd import trunk
    model = _model()
    batch = _batch()
    st = lambda: ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"])
    sem, _ = model(st(), is_train=True)
    assert type(sem.F.grad_fn).__name__ == "_TrunkFnBackward"
    with torch.no_grad():
        sem, _ = model(st(), is_train=True)
    assert sem.F.grad_fn is None
    model.eval()
    sem, _ = model(st())
    assert type(sem.F.grad_fn).__name__ != "_TrunkFnBackward"
    model.train()
    model.final.kernel.requires_grad_(False)
    sem, _ = model(st(), is_train=True)
    assert type(sem.F.grad_fn).__name__ != "_TrunkFnBackward"
    model.final.kernel.requires_grad_(True)
    # flat buffers: the pass that runs backward second finds its slices taken and accumulates through autograd
    from lidog_amd.optim import make_optimizer
    opt = make_optimizer("Adam", model, 1e-3)
    opt.zero_grad()
    a, _ = model(st(), is_train=True)
    b, _ = model(st(), is_train=True)
    assert type(a.F.grad_fn).__name__ == "_TrunkFnBackward" and type(b.F.grad_fn).__name__ == "_TrunkFnBackward"
    (a.F.square().mean() + b.F.square().mean()).backward()
    assert any(v is not None for v in _grads(model).values())
    # two executor passes against two operator-path passes on a fresh pair of models: the same two addends per parameter
    m1, m2 = _model(9), _model(9)
    o1, o2 = make_optimizer("Adam", m1, 1e-3), make_optimizer("Adam", m2, 1e-3)
    try:
        for m, o, on in ((m1, o1, True), (m2, o2, False)):
            trunk.set_enabled(on)
            o.zero_grad()
            a, _ = m(st(), is_train=True)
            b, _ = m(st(), is_train=True)
            (a.F.square().mean() + b.F.square().mean()).backward()
            o.flat.gather_strays()
        torch.cuda.synchronize()
    finally:
        trunk.set_enabled(True)
    g1, g2 = _grads(m1), _grads(m2)
    for k in g1:
        if g1[k] is None or g2[k] is None:
            assert g1[k] is None and g2[k] is None, k
        else:
            torch.testing.assert_close(g1[k], g2[k], rtol=1e-5, atol=1e-7, msg=lambda s: f"{k}: {s}")


def test_executor_with_bev_heads_on_every_decoder_level():
    """decoder_2d_level = all four levels (minkunet_bev.py:128-156): the executor hands back the three inner decoder
    levels as tensors of their own and takes their gradients; bit-identical to the operator path"""
    import lidog_amd
    import lidog_amd.me as ME
    from lidog_amd import trunk
    batch = _batch()
    res = {}
    try:
        for on in (False, True):
            trunk.set_enabled(on)
            m = lidog_amd.MinkUNet34BEV(1, 7, 3, decoder_2d_level=["bottle", "block6", "block7", "block8"],
                                        mapping_bound_2d=5.0).cuda()
            m.load_state_dict(seeded_state_dict(m, 11))
            m.train()
            sem, bev = m(ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"]),
                         is_train=True)
            assert sorted(bev) == ["block6", "block7", "block8", "bottle"]
            loss = sem.F.square().mean()
            for k in sorted(bev):
                loss = loss + bev[k].square().mean()
            loss.backward()
            torch.cuda.synchronize()
            res[on] = ({k: v.detach().clone() for k, v in bev.items()}, float(loss.detach()), _grads(m), _buffers(m))
    finally:
        trunk.set_enabled(True)
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], res[False][0][k]), k
    assert res[True][1] == res[False][1]
    _assert_same(res[True][2], res[False][2], "gradient")
    _assert_same(res[True][3], res[False][3], "buffer")


def test_maps_outlive_the_sparse_tensors():
    """the caller keeps nothing but the logits: the coordinate manager and its maps must stay alive (and untouched by
    whatever is allocated in between) until the executor's backward has run"""
    import gc
    import lidog_amd.me as ME
    from lidog_amd import trunk
    from lidog_amd.losses import SoftDICELoss
    batch = _batch()
    res = {}
    try:
        for on in (False, True):
            trunk.set_enabled(on)
            model = _model()
            logits = model(ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"]),
                           is_train=True)[0].F
            gc.collect()
            junk = [torch.full((1 << 20,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(64)]
            SoftDICELoss(ignore_label=-1)(logits, batch["source_sem_labels0"].long()).backward()
            torch.cuda.synchronize()
            del junk
            res[on] = _grads(model)
    finally:
        trunk.set_enabled(True)
    _assert_same(res[True], res[False], "gradient")


def test_executor_at_the_bench_shape_matches_the_operator_path():
    """one full-size batch (4 x 120 k points): executor and operator path, outputs and gradients bit for bit"""
    from lidog_amd import trunk
    from lidog_amd.train import SynthScans, build_model, build_step
    res = {}
    try:
        for on in (False, True):
            trunk.set_enabled(on)
            torch.manual_seed(0)
            model = build_model()
            model, step, _ = build_step(model)
            data = SynthScans(4)
            batch = data.batch([0, 1, 2, 3], torch.device("cuda"))
            out = step.training_step(batch)
            torch.cuda.synchronize()
            res[on] = ([float(out[k]) for k in ("loss", "sem_loss", "bev_loss")], _grads(model), _buffers(model))
            del model, step
            torch.cuda.empty_cache()
    finally:
        trunk.set_enabled(True)
    assert res[True][0] == res[False][0]
    _assert_same(res[True][1], res[False][1], "gradient")
    _assert_same(res[True][2], res[False][2], "state")


# ------------------------------------------------------------------ data parallel without SyncBatchNorm: hooks must fire
def _dp_plain_worker(rank, world, port, q):
    import os
    import sys
    import traceback
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from helpers import REPO
        sys.path.insert(0, REPO)
        torch.cuda.set_device(0)
        from lidog_amd import trunk
        from lidog_amd.optim import FlatAdam
        batch = _batch((81 + rank, 91 + rank))
        # (a) executor + gradient buckets of the process group (plain BatchNorm: the executor applies)
        model = _model()
        opt = FlatAdam(model, lr=1e-3, bucket_bytes=8 << 20)
        assert opt.buckets.active and opt.flat.hooked
        import lidog_amd.me as ME
        opt.zero_grad()
        sem, _ = model(ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"]),
                       is_train=True)
        assert type(sem.F.grad_fn).__name__ == "_TrunkFnBackward"
        sem.F.square().mean().backward()
        opt._prepare()
        torch.cuda.synchronize()
        # the buckets were counted down and reduced from C while backward was being queued; every gradient sits in
        # place in the flat buffer (no AccumulateGrad clone, no stray copy)
        assert opt.buckets.issued_early >= 1 and opt.strays == 0
        base = opt.flat.grad.data_ptr()
        for p, off in zip(opt.flat.params, opt.flat.offsets):
            assert p.grad is None or p.grad.data_ptr() == base + 4 * off
        got = opt.flat.grad.clone()
        # (b) operator path, no data parallelism, summed by hand
        trunk.set_enabled(False)
        ref = _model()
        ropt = FlatAdam(ref, lr=1e-3, local=True)
        ropt.zero_grad()
        sem, _ = ref(ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"]),
                     is_train=True)
        sem.F.square().mean().backward()
        ropt._prepare()
        torch.cuda.synchronize()
        want = ropt.flat.grad.clone()
        dist.all_reduce(want)
        ok = torch.equal(got, want)
        q.put((rank, ok, "" if ok else f"max |diff| {(got - want).abs().max().item():.3e}"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, False, f"{e!r}\n{traceback.format_exc()}"))
        raise


def test_executor_under_plain_data_parallel_fires_the_gradient_hooks():
    """DDP without SyncBatchNorm (local statistics): the executor runs and counts the gradient buckets down itself
    (lidog_trunk_backward, dp tables): flat gradient == sum over ranks of the operator path's local gradients, bit for bit"""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_plain_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    failed = not all(ok for _, ok, _ in got)
    for p in procs:
        p.join(10 if failed else 60)
        if p.is_alive():
            p.terminate()
            p.join(30)
    assert not failed, got


def test_second_backward_over_one_executor_pass_is_refused():
    """the executor reuses its activation arena once a pass has been backpropagated: a second backward over the same
    graph must raise instead of reading activations a later forward may have overwritten"""
    import lidog_amd.me as ME
    model = _model()
    b = _batch((65, 75))
    sem, _ = model(ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"]), is_train=True)
    assert type(sem.F.grad_fn).__name__ == "_TrunkFnBackward"
    loss = sem.F.square().mean()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already been backpropagated"):
        loss.backward()
