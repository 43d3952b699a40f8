"""Host side of the trunk executor (csrc/trunk.hip): the MinkUNet encoder-decoder as ONE autograd node.

The operator path (lidog_amd/me.py) walks the network module by module, as the reference does
(utils/models/minkunet_bev.py:302-374): ~550 autograd nodes and ~1 000 launches per training step, ~12 ms of python
per step.  The launch sequence is static, so it is written down ONCE per model as tables (`Program`) and executed
by `lidog_trunk_forward` / `lidog_trunk_backward` -- the same entry points in the same order with the same
arguments, hence bit-identical results (tests/test_gpu_trunk.py) -- behind a single `torch.autograd.Function`.

The executor takes the training step of a model whose trunk is built from the modules of lidog_amd.me, with local
BatchNorm statistics or -- a data-parallel rank (train_lidog.py:227-231) -- with SyncBatchNorm over ONE process group
and the optimiser's gradient buckets: the statistics all-reduces and the bucket all-reduces are then issued from C
(lidog_amd.comm).  Everything else (evaluation mode, no_grad, frozen parameters, LIDOG_TRUNK_EXEC=0) stays on the
operator path.
"""
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _lib, me as ME
from ._lib import call, call_on

ENABLED = os.environ.get("LIDOG_TRUNK_EXEC", "1") != "0"
if "LIDOG_TRUNK_FUSIONS" in os.environ:     # A/B runs: bit mask of the executor's fusions (see set_fusions)
    _lib.load().lidog_trunk_fusions(int(os.environ["LIDOG_TRUNK_FUSIONS"]))

KIND_K3, KIND_DOWN, KIND_UP, KIND_1X1, KIND_STEM = range(5)
OP_CONVBN, OP_CAT, OP_CONV = range(3)
TC_COLS, TM_COLS, TO_COLS, TB_COLS, REC_COLS = 20, 20, 8, 4, 4
(TC_KIND, TC_MAP, TC_CIN, TC_COUT, TC_K, TC_W, TC_WT, TC_GW, TC_BIAS, TC_GBIAS, TC_BNW, TC_BNB, TC_BNRM, TC_BNRV,
 TC_GBNW, TC_GBNB, TC_ITEMS, TC_NITEMS, TC_ITEMOFF) = range(19)
# external buffers: the input features and the tensors the model hands back
EXT_X, EXT_OUT, EXT_LOGITS, EXT_BOTTLE, EXT_LV_BOTTLE, EXT_LV_BLOCK6, EXT_LV_BLOCK7 = range(7)
N_EXT = 7
N_LEVELS = 5


def set_enabled(on):
    """trunk executor on / off (off: every step goes through the operator path)"""
    global ENABLED
    ENABLED = bool(on)


def set_fusions(mask):
    """which fusions the executor applies on top of the operator path's launch sequence (lidog_trunk_fusions; 1 =
    BatchNorm-backward statistics inside the producing data-gradient reduction, 2 = ReLU masks of the residual layers as
    bits, 4 = the BatchNorm + ReLU between the two convolutions of a block applied in the second one's staging); returns
    the previous mask"""
    return _lib.load().lidog_trunk_fusions(int(mask))


class _Unsupported(Exception):
    pass


def _kind_of(conv):
    if type(conv) is ME.MinkowskiConvolutionTranspose:
        if conv.kernel_size == 2 and conv.stride == 2 and conv.dilation == 1 and conv.in_channels % 4 == 0:
            return KIND_UP
        raise _Unsupported
    if type(conv) is not ME.MinkowskiConvolution or conv.dilation != 1:
        raise _Unsupported
    ks, st = conv.kernel_size, conv.stride
    if ks == 1 and st == 1:
        return KIND_1X1
    if ks == 2 and st == 2 and conv.in_channels % 4 == 0:
        return KIND_DOWN
    if ks % 2 == 1 and st == 1:
        if conv.in_channels == 1 and conv.out_channels in (16, 32, 64) and \
                conv.kernel_volume * conv.out_channels * 4 <= 48 * 1024:
            return KIND_STEM
        if conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0:
            return KIND_K3
    raise _Unsupported


class Program:
    """The static half: which convolution / BatchNorm runs on which buffers, in forward order."""

    def __init__(self, model):
        from .minkunet import _DEC, _ENC
        self.convs = []      # (conv module, BatchNorm module or None, kind, level in, level out)
        self.params = []     # every parameter of the trunk, in table order
        self.slots = []      # per convolution: indices into self.params of (kernel, bias, bn weight, bn bias)
        ops, bufs = [], []

        def buf(level, ch, ext=-1):
            bufs.append((level, ch, ext, 0))
            return len(bufs) - 1

        def conv(cv, bnm, in_b, level_in, level_out, relu=0, res=-1, fold=0, ext=-1):
            if bnm is not None and type(bnm) not in (ME.MinkowskiBatchNorm, ME.MinkowskiSyncBatchNorm):
                raise _Unsupported
            if bnm is not None and (cv.bias is not None or cv.out_channels % 4):
                raise _Unsupported
            self.convs.append((cv, bnm, _kind_of(cv), level_in, level_out))
            slot = [len(self.params), -1, -1, -1]
            self.params.append(cv.kernel)
            if cv.bias is not None:
                slot[1] = len(self.params)
                self.params.append(cv.bias)
            if bnm is not None:
                slot[2], slot[3] = len(self.params), len(self.params) + 1
                self.params += [bnm.bn.weight, bnm.bn.bias]
            self.slots.append(slot)
            out = buf(level_out, cv.out_channels, ext)
            ops.append((OP_CONVBN if bnm is not None else OP_CONV, len(self.convs) - 1, in_b, out, relu, res, fold, -1))
            return out

        def block(blk, in_b, level, ext=-1):
            if type(blk) is not ME.BasicBlock or blk.conv1.stride != 1:
                raise _Unsupported
            h = conv(blk.conv1, blk.norm1, in_b, level, level, relu=1, fold=1)
            res = in_b
            if blk.downsample is not None:
                ds = blk.downsample
                if not (isinstance(ds, torch.nn.Sequential) and len(ds) == 2):
                    raise _Unsupported
                res = conv(ds[0], ds[1], in_b, level, level)
            return conv(blk.conv2, blk.norm2, h, level, level, relu=1, res=res, ext=ext)

        def stage(blocks, in_b, level, ext=-1):
            blocks = list(blocks)
            for i, blk in enumerate(blocks):
                in_b = block(blk, in_b, level, ext if i == len(blocks) - 1 else -1)
            return in_b

        x = buf(0, model.conv0p1s1.in_channels, EXT_X)
        out = conv(model.conv0p1s1, model.bn0, x, 0, 0, relu=1)
        skips = [out]
        for idx, (i, s) in enumerate(_ENC):
            out = conv(getattr(model, f"conv{i}p{s}s2"), getattr(model, f"bn{i}"), out, idx, idx + 1, relu=1)
            out = stage(getattr(model, f"block{i}"), out, idx + 1, EXT_BOTTLE if idx == len(_ENC) - 1 else -1)
            skips.append(out)
        skips.pop()
        level_ext = [EXT_LV_BOTTLE, EXT_LV_BLOCK6, EXT_LV_BLOCK7, EXT_OUT]
        level = len(_ENC)
        for (j, s), ext in zip(_DEC, level_ext):
            out = conv(getattr(model, f"convtr{j}p{s}s2"), getattr(model, f"bntr{j}"), out, level, level - 1, relu=1)
            level -= 1
            skip = skips.pop()
            cat = buf(level, bufs[out][1] + bufs[skip][1])
            ops.append((OP_CAT, -1, out, cat, 0, -1, 0, skip))
            out = stage(getattr(model, f"block{j + 1}"), cat, level, ext)
        if level != 0 or model.final.bias is None:
            raise _Unsupported
        conv(model.final, None, out, 0, 0, ext=EXT_LOGITS)
        self.ops = np.array(ops, dtype=np.int64)
        self.bufs = np.array(bufs, dtype=np.int64)
        self.ext_shape = {}
        for lv, ch, ext, _ in bufs:
            if ext > 0:
                self.ext_shape[ext] = (lv, ch)
        # kernel-map keys (me.CoordinateManager) and the trace of map uses in forward order
        self.map_keys, self.conv_map, self.trace = [], [], []
        for cv, _, kind, l_in, l_out in self.convs:
            s_in, s_out = 2 ** l_in, 2 ** l_out
            if kind == KIND_1X1:
                key = ("identity", s_in)
            elif kind == KIND_UP:
                key = (s_out, s_in, cv.kernel_size, cv.dilation)
            else:
                key = (s_in, s_out, cv.kernel_size, cv.dilation)
            if key not in self.map_keys:
                self.map_keys.append(key)
            self.conv_map.append(self.map_keys.index(key))
            self.trace.append((key, cv.in_channels, cv.out_channels))
        self.bns = [bnm for _, bnm, _, _, _ in self.convs if bnm is not None]
        # which row lists every map needs
        self.rows_need = [set() for _ in self.map_keys]
        # os_ok[mi]: every 3^3 convolution on the map has channel counts the output-stationary kernel takes; where the
        # map's rows are sorted for it (me.KernelMap.sorted) nobody walks its per-row lists and they are not built
        self.os_ok = [True for _ in self.map_keys]
        for (cv, _, kind, _, _), mi in zip(self.convs, self.conv_map):
            if kind == KIND_K3:
                self.rows_need[mi] |= {"out", "in"}
                if cv.in_channels % 32 or cv.out_channels % 32:
                    self.os_ok[mi] = False
            elif kind in (KIND_DOWN, KIND_UP):
                self.rows_need[mi].add("out")
        self.n_rec = len(ops) * REC_COLS + len(bufs)


class _Arenas:
    """Device memory of the executor, kept per model and grown on demand (the caching allocator would hand the same
    blocks back every step anyway; keeping them skips ~10 allocator calls per step and any fragmentation)."""

    def __init__(self):
        self.bufs = {}
        self.fwd_owner = None

    def get(self, name, nbytes, device):
        t = self.bufs.get(name)
        if t is None or t.numel() < nbytes or t.device != device:
            t = torch.empty(int(nbytes * 1.1) + 4096, dtype=torch.uint8, device=device)
            self.bufs[name] = t
        return t


def _structure(model):
    """identity of every module and parameter of the model: the program holds references to them, so a replaced layer
    (a new classifier, convert_sync_batchnorm after the first step, a re-assigned parameter) must rebuild it"""
    return tuple(id(m) for m in model.modules()), tuple(id(p) for p in model.parameters())


# Walking the model costs ~0.3 ms: it is done again only after SOME module of the process registered a submodule or a
# parameter (torch's global registration hooks fire on add_module / register_parameter / attribute assignment).
_EPOCH = [0]


def _bump(*_):
    _EPOCH[0] += 1


torch.nn.modules.module.register_module_module_registration_hook(_bump)
torch.nn.modules.module.register_module_parameter_registration_hook(_bump)


def program_of(model):
    cached = model.__dict__.get("_lidog_trunk_program")
    if cached is not None and cached[2] == _EPOCH[0]:
        return cached[1]
    sig = _structure(model)
    if cached is None or cached[0] != sig:
        try:
            prog = Program(model)
        except (_Unsupported, AttributeError):
            prog = None
        cached = (sig, prog, _EPOCH[0])
        model.__dict__.setdefault("_lidog_trunk_arenas", _Arenas())
    else:
        cached = (cached[0], cached[1], _EPOCH[0])
    model.__dict__["_lidog_trunk_program"] = cached
    return cached[1]


class _Run:
    """one forward pass through the executor: tables, arena offsets and what backward needs"""
    done = False


def _addr(t):
    return 0 if t is None else t.data_ptr()


def _eligible(model, prog, x):
    if not (ENABLED and prog is not None and torch.is_grad_enabled() and model.training):
        return False
    f = x.F
    if not (f.is_cuda and f.dtype == torch.float32 and not f.requires_grad and x.coordinate_map_key == 1):
        return False
    group = prog.bns[0]._sync_group() if prog.bns else None
    for bnm in prog.bns:
        bn = bnm.bn
        if not (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None) or \
                bnm._sync_group() is not group:
            return False
    for p in prog.params:
        if not p.requires_grad or p.dtype != torch.float32 or not p.is_cuda:
            return False
    return True


def _static_rows(prog):
    """the columns of the convolution table that only change when the parameters move in memory (a new optimiser
    re-homes them into its flat buffers, .to(device), ...): rebuilt when any parameter's address differs"""
    sig = tuple(p.data_ptr() for p in prog.params)
    cached = prog.__dict__.get("_static")
    if cached is not None and cached[0] == sig:
        return cached[1]
    rows = np.zeros((len(prog.convs), TC_COLS), dtype=np.int64)
    for i, (cv, bnm, kind, _, _) in enumerate(prog.convs):
        K = 1 if kind == KIND_1X1 else cv.kernel_volume
        row = [kind, prog.conv_map[i], cv.in_channels, cv.out_channels, K, cv.kernel.data_ptr(), 0, 0, _addr(cv.bias)]
        rows[i, :len(row)] = row
        if bnm is not None:
            bn = bnm.bn
            rows[i, TC_BNW:TC_BNRV + 1] = [bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                           bn.running_var.data_ptr()]
    prog._static = (sig, rows)
    return rows


def _flat_targets(prog):
    """(owner, gradient-pointer columns [n_convs, 4], persistent views into the flat gradient buffer) when EVERY
    parameter of the trunk lives in one optimiser's flat buffers (lidog_amd.optim.FlatParams), else None"""
    ref0 = getattr(prog.params[0], "_flat_ref", None)
    if ref0 is None:
        return None
    cached = prog.__dict__.get("_flat")
    if cached is not None and cached[0] is ref0[2] and cached[3] == ref0[0].data_ptr():
        return cached
    views = []
    for p in prog.params:
        ref = getattr(p, "_flat_ref", None)
        if ref is None or ref[2] is not ref0[2]:
            return None
        views.append(ref[0][ref[1]:ref[1] + p.numel()].view(p.shape))
    cols = np.zeros((len(prog.convs), 4), dtype=np.int64)
    for ci, slot in enumerate(prog.slots):
        cols[ci] = [views[s].data_ptr() if s >= 0 else 0 for s in slot]
    prog._flat = (ref0[2], cols, views, ref0[0].data_ptr())
    return prog._flat


def _build_tables(prog, x, run):
    """per-batch tables; None when a map of this batch is degenerate (no pairs): operator path"""
    cm = x.coordinate_manager
    maps = np.zeros((len(prog.map_keys), TM_COLS), dtype=np.int64)
    keep = []
    mobjs = []
    for i, key in enumerate(prog.map_keys):
        if key[0] == "identity":
            if key[1] not in cm.maps:
                return None
            m = cm.identity_map(cm.maps[key[1]].n)
            row = [1, m.n_in, m.n_out, m.P, 0, 0, 0, 0, 0, 0, _addr(m.tiles), m.n_tiles, 0, _addr(m.rows)]
        else:
            m = cm.kernel_map(*key)
            rp_o = rl_o = rp_i = rl_i = None
            srt = m.sorted() or (None, None, None)   # sorted rows: the output-stationary 3^3 kernel applies (me.KernelMap.sorted)
            by_rows = srt[0] is None or not prog.os_ok[i] or _lib.load().lidog_get_sparse_core() != 1
            if "out" in prog.rows_need[i] and by_rows:
                rp_o, rl_o = m.rows("out")
            if "in" in prog.rows_need[i] and by_rows:
                rp_i, rl_i = m.rows("in")
            row = [m.K, m.n_in, m.n_out, m.P, _addr(m.pair_in), _addr(m.pair_out), _addr(rp_o), _addr(rl_o),
                   _addr(rp_i), _addr(rl_i), _addr(m.tiles), m.n_tiles, _addr(m.nbr), 0, _addr(srt[0]), _addr(srt[1]),
                   _addr(srt[2])]
        if m.P == 0 or m.n_tiles == 0 or not m.tiles.is_contiguous():
            return None
        maps[i, :len(row)] = row
        mobjs.append(m)
    levels = []
    for lv in range(N_LEVELS):
        if 2 ** lv not in cm.maps:
            return None
        if cm.maps[2 ** lv].n == 0:
            return None
        levels.append(cm.maps[2 ** lv].n)
    for i, (cv, bnm, kind, _, _) in enumerate(prog.convs):
        if kind == KIND_STEM and mobjs[prog.conv_map[i]].nbr is None:
            return None
    # from here on the pass is taken: the BatchNorm batch counters move (me._training_momentum)
    convs = _static_rows(prog).copy()
    conv_f = np.empty((len(prog.convs), 2), dtype=np.float64)
    dyn = []
    for i, (cv, bnm, kind, _, _) in enumerate(prog.convs):
        m = mobjs[prog.conv_map[i]]
        w = cv.kernel
        wt = w._wt_view.data_ptr() if getattr(w, "_wt_version", -2) == w._version else 0
        items, n_items, item_off = ME._wgrad_items(m, cv.in_channels, cv.out_channels)
        keep.append((items, item_off))
        dyn.append((wt, items.data_ptr(), n_items, item_off.data_ptr()))
        conv_f[i] = (bnm.bn.eps, ME._training_momentum(bnm.bn)) if bnm is not None else (0.0, 0.0)
    convs[:, [TC_WT, TC_ITEMS, TC_NITEMS, TC_ITEMOFF]] = np.array(dyn, dtype=np.int64)
    # the tables hold raw device addresses: the maps they point into must outlive the autograd node (the caller may
    # drop every SparseTensor, and with it the coordinate manager, right after taking the features)
    run.cm, run.map_objects = cm, mobjs
    run.maps, run.convs, run.conv_f, run.keep = maps, convs, conv_f, keep
    run.levels = np.array(levels, dtype=np.int64)
    run.level_list = levels
    return run


DP_COLS = 12
(DP_SYNC_BN, DP_COMM_BN, DP_CALLBACK, DP_COMM_GRAD, DP_COMM_STREAM, DP_GRAD_BASE, DP_N_BUCKETS, DP_BUCKETS, DP_PENDING,
 DP_PARAM_BUCKET, DP_PEER) = range(11)
_CALLBACK_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64)


class _Collectives:
    """The `dp` argument of lidog_trunk_forward / _backward for one pass (include/lidog_amd.h): SyncBatchNorm statistics
    over `group` (None = local statistics) and/or the gradient buckets of the optimiser that owns the trunk's
    parameters.  Native transport: communicator handles; torch transport: a host callback into torch.distributed."""

    def __init__(self, run, group):
        from .comm import transport
        self.run, self.group = run, group
        self.error = None
        self.tr = transport(group) if group is not None else None
        self.desc = np.zeros(DP_COLS, dtype=np.int64)
        self.cb = _CALLBACK_T(self._callback)     # kept alive with the run
        self.desc[DP_CALLBACK] = ctypes.cast(self.cb, ctypes.c_void_p).value
        if self.tr is not None:
            self.desc[DP_SYNC_BN] = 1
            if self.tr.kind == "native":
                self.desc[DP_COMM_BN] = self.tr.comm_bn
            if self.tr.peer:
                self.desc[DP_PEER] = self.tr.peer
        self.regions = ()         # the tensors the statistics messages live in (set per call)
        self.buckets = None

    def use_buckets(self, buckets, prog):
        """hand the countdown of the gradient buckets to C for this backward pass"""
        tr = buckets.transport
        self.buckets = buckets
        d = self.desc
        cached = prog.__dict__.get("_param_bucket")
        if cached is None or cached[0] is not buckets:
            by_slot = [[prog.params[s] if s >= 0 else None for s in slot] for slot in prog.slots]
            cached = prog._param_bucket = (buckets, buckets.executor_tables(by_slot))
        self.param_bucket = cached[1]
        d[DP_N_BUCKETS] = len(buckets.slices)
        d[DP_BUCKETS] = buckets.slice_table.ctypes.data
        d[DP_PENDING] = buckets.pending.ctypes.data
        d[DP_PARAM_BUCKET] = self.param_bucket.ctypes.data
        if tr.bucket_kind == "native":
            d[DP_COMM_GRAD], d[DP_COMM_STREAM] = tr.comm_grad, tr.raw_stream
            d[DP_GRAD_BASE] = buckets.flat.grad.data_ptr()
        else:
            d[DP_COMM_GRAD] = d[DP_COMM_STREAM] = d[DP_GRAD_BASE] = 0

    def no_buckets(self):
        self.buckets = None
        self.desc[DP_N_BUCKETS] = 0

    def _callback(self, what, a, b):
        try:
            if what == 0:    # all-reduce of b doubles at device address a (inside one of the regions of this call)
                for region in self.regions:
                    off = a - region.data_ptr()
                    if 0 <= off and off + 8 * b <= region.numel():
                        self.tr.allreduce_f64(region[off:off + 8 * b].view(torch.float64))
                        break
                else:
                    raise RuntimeError("statistics message outside the regions of this call")
            elif what == 1:
                # the bucket's weight gradients may still be running on the lane stream: mark it busy so that the
                # reduction is ordered behind it (GradientBuckets._reduce), until the optimiser's join
                dev = self.buckets.flat.grad.device
                if ME._WgradLane.active():
                    ME._WgradLane.get(dev).pending = True
                self.buckets.issued_early += 1
                self.buckets._reduce(int(a))
            else:
                raise RuntimeError(f"unknown collective request {what}")
            return 0
        except BaseException as exc:   # noqa: BLE001 -- must not propagate through the C frames
            self.error = exc
            return 1

    def ptr(self):
        return self.desc.ctypes.data

    def check(self):
        if self.error is not None:
            err, self.error = self.error, None
            raise err


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, run, *params):
        prog, dev = run.prog, feats.device
        outs = [None] * N_EXT
        outs[EXT_X] = feats
        for ext, (lv, ch) in prog.ext_shape.items():
            outs[ext] = torch.empty((run.level_list[lv], ch), dtype=torch.float32, device=dev)
        run.ext = np.array([t.data_ptr() for t in outs], dtype=np.int64)
        run.rec = np.zeros(prog.n_rec, dtype=np.int64)
        need = np.zeros(2, dtype=np.int64)
        args = (run.convs.ctypes.data, run.conv_f.ctypes.data, len(prog.convs), run.maps.ctypes.data, len(run.maps),
                prog.ops.ctypes.data, len(prog.ops), prog.bufs.ctypes.data, len(prog.bufs), run.levels.ctypes.data,
                run.ext.ctypes.data)
        coll = run.coll
        dp = coll.ptr() if coll is not None else None
        # the second stream of the backward pass is idle in the forward pass: the downsample branches of the layers' first
        # blocks run on it (csrc/trunk.hip); `call_on` appends it behind the launch stream
        lane = ME._WgradLane.get(dev) if ME._WgradLane.active() else None
        lane_raw = lane.raw if lane is not None else None
        call_on(lane_raw, "lidog_trunk_forward", *args, None, 0, None, 0, run.rec.ctypes.data, need.ctypes.data, 1, dp,
                _lib.stream())
        arenas = run.arenas
        owner = arenas.fwd_owner() if arenas.fwd_owner is not None else None
        if owner is not None and not owner.done:
            # the previous pass has not seen its backward yet: its activations stay where they are
            arena = torch.empty(int(need[0]) + 4096, dtype=torch.uint8, device=dev)
        else:
            arena = arenas.get("fwd", int(need[0]), dev)
            arenas.fwd_owner = weakref.ref(run)
        scratch = arenas.get("scratch", int(need[1]), dev)
        if coll is not None:
            coll.regions = (scratch, arena)
        try:
            call_on(lane_raw, "lidog_trunk_forward", *args, arena.data_ptr(), arena.numel(), scratch.data_ptr(),
                    scratch.numel(), run.rec.ctypes.data, need.ctypes.data, 0, dp, _lib.stream())
        except RuntimeError:
            if coll is not None:
                coll.check()
            raise
        run.args, run.arena = args, arena
        ctx.run = run
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*outs)
        return tuple(outs[1:])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        run = ctx.run
        if run.done:
            # the activation arena of this pass may already belong to a later forward pass (it is reused as soon as a
            # backward pass has run): a second backward over the same graph (retain_graph=True, two losses
            # backpropagated one after the other) would silently read overwritten activations
            raise RuntimeError("lidog_amd.trunk: this forward pass has already been backpropagated once; call the model "
                               "again (or set LIDOG_TRUNK_EXEC=0 to use the operator path, which keeps its tensors)")
        prog = run.prog
        outs = ctx.saved_tensors
        dev = outs[0].device
        gouts = [None] + [g.contiguous() if g is not None else None for g in gouts]
        ext_grad = np.array([_addr(g) for g in gouts], dtype=np.int64)
        # where the parameter gradients go: the optimiser's flat buffer (me._grad_out) when this pass owns the slice,
        # else one fresh buffer for the rest
        convs = run.convs
        # A slice of the flat buffer is handed out at most once per gradient generation (me._grad_out): when the
        # model is called twice before one backward pass (trainer_lighting_2d_multi.py:166-167) the pass that runs
        # second computes into fresh tensors and autograd accumulates; this call has joined the lane by then.
        flat = _flat_targets(prog)
        direct = False
        if flat is not None:
            gen = flat[0].generation
            direct = all(p.grad is None and p._flat_taken != gen for p in prog.params)
            if direct:
                for p in prog.params:
                    p._flat_taken = gen
        if direct:
            # the common case: every gradient goes straight into the flat buffer through views made once; they are
            # bound to .grad below, by hand (188 AccumulateGrad nodes would only do the same assignment)
            views = flat[2]
            convs[:, [TC_GW, TC_GBIAS, TC_GBNW, TC_GBNB]] = flat[1]
        else:
            views, fresh = [None] * len(prog.params), []
            for i, p in enumerate(prog.params):
                ref = getattr(p, "_flat_ref", None)
                if ref is not None and p.grad is None and p._flat_taken != ref[2].generation:
                    p._flat_taken = ref[2].generation
                    views[i] = ref[0][ref[1]:ref[1] + p.numel()].view(p.shape)
                else:
                    fresh.append(i)
            if fresh:
                pool = torch.empty(sum(prog.params[i].numel() for i in fresh), dtype=torch.float32, device=dev)
                off = 0
                for i in fresh:
                    n = prog.params[i].numel()
                    views[i] = pool[off:off + n].view(prog.params[i].shape)
                    off += n
            for ci, slot in enumerate(prog.slots):
                convs[ci, TC_GW] = views[slot[0]].data_ptr()
                if slot[1] >= 0:
                    convs[ci, TC_GBIAS] = views[slot[1]].data_ptr()
                if slot[2] >= 0:
                    convs[ci, TC_GBNW] = views[slot[2]].data_ptr()
                    convs[ci, TC_GBNB] = views[slot[3]].data_ptr()
        lane = ME._WgradLane.get(dev) if ME._WgradLane.active() else None
        need = np.zeros(3, dtype=np.int64)
        done = np.zeros(len(prog.convs), dtype=np.int32)
        # Data-parallel gradient buckets (lidog_amd.optim.GradientBuckets): with every gradient going straight into the
        # flat buffer the countdown of the trunk's parameters runs in C and a bucket is reduced as soon as its last
        # gradient is queued; otherwise the gradients go back through autograd and the parameters' hooks count.
        buckets = flat[0].buckets if (direct and flat[0].buckets is not None and flat[0].buckets.active) else None
        coll = run.coll
        if buckets is not None and coll is None:
            coll = run.coll = _Collectives(run, None)
        if coll is not None:
            if buckets is not None:
                coll.use_buckets(buckets, prog)
            else:
                coll.no_buckets()
        dp = (coll.ptr() if coll is not None else None,)
        args = run.args + (ext_grad.ctypes.data, run.arena.data_ptr(), run.rec.ctypes.data)
        tail = (need.ctypes.data, done.ctypes.data)
        mode = (1 if (lane is not None and ME._WgradLane.mode == 1) else 0,)
        lane_raw = lane.raw if lane is not None else None
        call_on(lane_raw, "lidog_trunk_backward", *args, None, 0, None, 0, None, 0, *tail, 1, *mode, *dp, _lib.stream())
        arenas = run.arenas
        garena = arenas.get("grad", int(need[0]), dev)
        scratch = arenas.get("scratch", int(need[1]), dev)
        lscratch = arenas.get("lane", max(int(need[2]), 256), dev)
        if coll is not None:
            coll.regions = (scratch, garena)
        open_before = buckets.pending > 0 if buckets is not None else None
        try:
            call_on(lane_raw, "lidog_trunk_backward", *args, garena.data_ptr(), garena.numel(), scratch.data_ptr(),
                    scratch.numel(), lscratch.data_ptr(), lscratch.numel(), *tail, 0, *mode, *dp, _lib.stream())
        except RuntimeError:
            if coll is not None:
                coll.check()
            raise
        if buckets is not None and buckets.transport.bucket_kind == "native":
            buckets.issued_early += int((open_before & (buckets.pending == 0)).sum())
        run.done = True
        grads = [None] * len(prog.params)
        params = prog.params
        # without bucket tables in C the gradients of a hooked optimiser go back through autograd so that the hooks fire
        bind = direct and (buckets is not None or not flat[0].hooked)
        for ci, slot in enumerate(prog.slots):
            if done[ci]:
                for s in slot:
                    if s >= 0:
                        if bind:
                            params[s].grad = views[s]
                        else:
                            grads[s] = views[s]
        return (None, None, *grads)


def trunk_forward(model, x):
    """(out, bottle, levels, logits) of `_Trunk._trunk_forward` + classifier through the executor, or None when this
    call has to take the operator path"""
    prog = program_of(model)
    if not _eligible(model, prog, x):
        return None
    run = _Run()
    run.prog, run.arenas = prog, model.__dict__["_lidog_trunk_arenas"]
    group = prog.bns[0]._sync_group() if prog.bns else None
    if _build_tables(prog, x, run) is None:
        return None
    run.coll = _Collectives(run, group) if group is not None else None
    cm = x.coordinate_manager
    cm.trace.extend(prog.trace)
    feats = x.F.contiguous()
    out, logits, bottle, lv_bottle, lv6, lv7 = _TrunkFn.apply(feats, run, *prog.params)

    def st(t, level):
        return ME.SparseTensor(t, coordinate_manager=cm, coordinate_map_key=2 ** level)

    levels = {"bottle": st(lv_bottle, 3), "block6": st(lv6, 2), "block7": st(lv7, 1), "block8": st(out, 0)}
    return levels["block8"], st(bottle, 4), levels, st(logits, 0)
