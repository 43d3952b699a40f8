"""Optimisers, learning-rate schedules and the gradient all-reduce of the LiDOG step on flat HBM buffers.

Restates what the reference configures through pytorch-lightning (python the GPU box never receives):
  configure_optimizers  utils/pipelines/trainer_lighting_2d.py:349-394, utils/pipelines/trainer_lighting.py:335-380
      Adam(lr, weight_decay=1e-4)  |  SGD(lr, momentum=0.98, weight_decay=1e-4, nesterov=True)      (:26-27,351-360)
      CosineAnnealingLR(T_max=10) | ExponentialLR(gamma=0.99) | CyclicLR(lr/1e4 .. lr, step_size_up=5,
      mode="triangular2", cycle_momentum=False)                                                        (:379-389)
      returned as ([optimizer], [scheduler]) => Lightning steps the scheduler once per training EPOCH
  DDP gradient averaging  train_lidog.py:227-231 (strategy='ddp')

Parameters, gradients and optimiser state live in contiguous fp32 buffers: one fused HIP kernel per step
(csrc/conv2d.hip:k_adam, csrc/optim.hip:k_sgd), one RCCL all-reduce per 32 MiB bucket.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import me as ME
from ._lib import call, call_on, ptr


class FlatParams:
    """Re-homes every parameter (and its .grad) of `model` into two contiguous fp32 buffers."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        self.generation = 0
        self.hooked = False
        self.buckets = None    # the GradientBuckets that reduce this buffer (data-parallel runs)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            # lets backward kernels write this gradient in place (me._grad_out): (buffer, offset, owner)
            p._flat_ref = (self.grad, off, self)
            p._flat_taken = -1
            self.offsets.append(off)
            off += n
        self.total = total

    def zero_grad(self):
        """set_to_none semantics: backward kernels write fresh views of the flat buffer and autograd adopts them
        as .grad, so there is no `grad += new` pass; the buffer is cleared for parameters that get no gradient.
        A new generation starts: every parameter's slice may be handed out (once) again (me._grad_out)."""
        self.grad.zero_()
        self.generation += 1
        for p in self.params:
            p.grad = None

    def in_place(self, p, off):
        return p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * off

    def gather_strays(self):
        """gradients that autograd produced outside the flat buffer (e.g. a bias gradient from a torch op) are
        copied in with one fused call; returns how many there were"""
        dst, src = [], []
        base = self.grad.data_ptr()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.data_ptr() != base + 4 * off:
                view = self.grad[off:off + p.numel()].view(p.shape)
                dst.append(view)
                src.append(p.grad)
                p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)
        return len(dst)

    def broadcast(self, group=None, src=0):
        """DDP's start-up broadcast: every rank starts from rank `src`'s parameters"""
        dist.broadcast(self.flat, src=src, group=group)


class GradientBuckets:
    """Bucketed all-reduce (sum) of the flat gradient buffer, overlapped with backward.

    Parameters are registered in forward order, gradients arrive roughly in reverse, so buckets are
    contiguous slices walked from the END of the buffer.  xGMI is point-to-point (7 links per GPU):
    a few large messages (default 32 MiB) keep every link busy without paying per-message latency.

    A bucket is reduced as soon as its last gradient has been produced.  With the second backward stream active
    (me._WgradLane) the bucket's weight gradients may still be running there: the collective is issued from THAT
    stream (RCCL's stream then waits for the lane, which itself waits for everything queued on the main stream so far)
    and the main stream never waits before finish().
    RCCL executes the collectives of ONE communicator in issue order on one stream, so a bucket that still waits for
    the lane holds back the SyncBatchNorm statistics all-reduces issued after it -- by at most the lane's lag of about
    one layer.  `own_communicator=True` gives the buckets a communicator of their own (`dist.new_group`) and removes
    that coupling; it is not the default because two communicators in flight at once could not be exercised on more
    than one rank in this build's environment, and a same-communicator schedule is identical on every rank by
    construction."""

    single_rank = False   # test hook: bucket and all-reduce even in a one-rank process group

    def __init__(self, flat, group=None, bucket_bytes=32 << 20, own_communicator=None, local=False):
        """`local=True`: no data parallelism for this optimiser even inside an initialised process group"""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if (dist.is_initialized() and not local) else 1
        self.active = self.world > 1 or (self.single_rank and dist.is_initialized() and not local)
        self.handles = []
        self.index_of = {id(p): i for i, p in enumerate(flat.params)}
        self.bucket_of = {}
        self.pending0 = []
        self.slices = []
        self.issued_early = 0     # buckets reduced while backward was still being queued (hook or trunk executor)
        self.transport = None
        # LIDOG_DP_SAFE=1 (bench.py's fallback after a hung N > 1 run): no bucket leaves before backward has ended; finish()
        # reduces them one after the other on the compute stream through torch.distributed -- no bucket stream, no
        # second communicator in flight next to the statistics messages
        import os
        self.deferred = os.environ.get("LIDOG_DP_SAFE") == "1"
        if not self.active:
            return
        from .comm import transport
        self.transport = transport(group)
        if own_communicator and group is None:
            self.group = dist.new_group(ranks=list(range(dist.get_world_size())))
        cur_lo = cur_hi = flat.total
        count = 0
        members = []
        for p, off in reversed(list(zip(flat.params, flat.offsets))):
            members.append(p)
            cur_lo = off
            count += 1
            if (cur_hi - cur_lo) * 4 >= bucket_bytes:
                self._close(members, cur_lo, cur_hi, count)
                members, count, cur_hi = [], 0, cur_lo
        if members:
            self._close(members, cur_lo, cur_hi, count)
        # countdown per bucket, shared with the trunk executor (csrc/trunk.hip counts the trunk's parameters down in C
        # and reduces a bucket itself when it reaches 0): a host int32 array, not a list
        self.pending = np.array(self.pending0, dtype=np.int32)
        self.slice_table = np.array(self.slices, dtype=np.int64).reshape(-1, 2)
        for p in flat.params:
            p.register_post_accumulate_grad_hook(self._hook)
        flat.hooked = True     # parameters outside the trunk executor reach their bucket through these hooks
        flat.buckets = self

    def _close(self, members, lo, hi, count):
        b = len(self.slices)
        self.slices.append((lo, hi))
        self.pending0.append(count)
        for p in members:
            self.bucket_of[id(p)] = b

    def _hook(self, p):
        off = self.flat.offsets[self.index_of[id(p)]]
        if not self.flat.in_place(p, off):   # stray gradient: bring it into the flat buffer before it is reduced
            view = self.flat.grad[off:off + p.numel()].view(p.shape)
            view.copy_(p.grad)
            p.grad = view
        b = self.bucket_of[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self.issued_early += 1
            self._reduce(b)

    def _reduce(self, b):
        if self.deferred:      # finish() reduces every bucket
            return
        lo, hi = self.slices[b]
        buf = self.flat.grad[lo:hi]
        lane = ME.wgrad_lane(buf.device) if buf.is_cuda else None
        tr = self.transport
        if tr.bucket_kind == "native":
            # on the bucket stream, behind what the compute stream and the weight-gradient stream hold now; nothing
            # waits for it before finish()
            tr.stream.wait_stream(torch.cuda.current_stream(buf.device))
            if lane is not None:
                tr.stream.wait_stream(lane.stream)
            call_on(tr.raw_stream, "lidog_allreduce_f32", ptr(buf), hi - lo, tr.comm_grad)
            return
        if lane is not None:
            main = torch.cuda.current_stream(buf.device)
            lane.stream.wait_stream(main)          # BatchNorm / bias gradients of the bucket are written on main
            with torch.cuda.stream(lane.stream):
                self.handles.append(dist.all_reduce(buf, group=self.group, async_op=True))
            return
        self.handles.append(dist.all_reduce(buf, group=self.group, async_op=True))

    def finish(self):
        """wait for every bucket; buckets whose hooks did not all fire (unused parameters) are reduced now"""
        if not self.active:
            return
        if self.deferred:
            # the caller (_FlatOptimizer._prepare) has joined the weight-gradient stream: every gradient is complete on
            # the current stream
            for lo, hi in self.slices:
                dist.all_reduce(self.flat.grad[lo:hi], group=self.group)
            self.pending[:] = self.pending0
            return
        for b, left in enumerate(self.pending.tolist()):
            if left > 0:
                self._reduce(b)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.transport.bucket_kind == "native":
            torch.cuda.current_stream(self.flat.grad.device).wait_stream(self.transport.stream)
        self.pending[:] = self.pending0

    def executor_tables(self, params_by_slot):
        """int32 [n, 4] bucket of each (kernel, bias, BatchNorm weight, BatchNorm bias) of the trunk's convolutions,
        -1 where a convolution has no such parameter (lidog_trunk_backward, dp[9])"""
        out = np.full((len(params_by_slot), 4), -1, dtype=np.int32)
        for ci, slot in enumerate(params_by_slot):
            for j, p in enumerate(slot):
                if p is not None:
                    out[ci, j] = self.bucket_of[id(p)]
        return out


class TransposedKernels:
    """[K, Cout, Cin] copies of every sparse-convolution kernel [K, Cin, Cout] of the model (what the data gradients of
    the backward pass multiply by), refreshed by ONE launch after each optimiser step instead of one small transpose
    kernel per convolution inside the backward pass's dependent chain (63 launches per step).  A copy is used only
    while the parameter's version counter still has the value it had at the refresh: weights changed by anything
    else (load_state_dict, an in-place edit) fall back to the per-layer transpose (me._SparseConvFn.backward)."""

    def __init__(self, flat):
        self.flat = flat
        self.items = []
        desc, off, tiles = [], 0, 0
        for p, src in zip(flat.params, flat.offsets):
            if not getattr(p, "_lidog_sparse_kernel", False):
                continue
            shape = p.shape if p.dim() == 3 else (1,) + tuple(p.shape)
            K, Cin, Cout = shape
            desc.append((src, off, K, Cin, Cout, tiles))
            self.items.append((p, off, (K, Cout, Cin)))
            off += p.numel()
            tiles += K * (-(-Cin // 32)) * (-(-Cout // 32))
        self.total_tiles = tiles
        dev = flat.flat.device
        self.buf = torch.empty(off, dtype=torch.float32, device=dev)
        self.desc = torch.tensor(desc, dtype=torch.int64, device=dev).view(-1, 6) if desc else None
        for p, o, shape in self.items:
            p._wt_view = self.buf[o:o + p.numel()].view(shape)
            p._wt_version = -1
        self.refresh()

    def refresh(self):
        if not self.items or not self.buf.is_cuda:
            return
        call("lidog_transpose_batched", ptr(self.flat.flat), ptr(self.buf), ptr(self.desc), len(self.items),
             self.total_tiles)
        for p, _, _ in self.items:
            p._wt_version = p._version


class _FlatOptimizer:
    """Common part of the fused optimisers: flat buffers, gradient buckets, and torch's rule that a parameter
    WITHOUT a gradient is skipped entirely (no weight decay, no moment decay, no step count) -- e.g. `final.*`
    while `epoch < warmup_epochs` (trainer_lighting_2d.py:193-201)."""

    def __init__(self, model, lr, group=None, bucket_bytes=32 << 20, local=False):
        self.flat = FlatParams(model)
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.param_steps = [0] * len(self.flat.params)   # torch keeps `step` per parameter
        self.buckets = GradientBuckets(self.flat, group, bucket_bytes, local=local)
        if self.buckets.world > 1:
            self.flat.broadcast(group)     # DDP's start-up broadcast of rank 0's parameters
        self.strays = 0
        self.transposed = TransposedKernels(self.flat)

    @property
    def steps(self):
        return max(self.param_steps) if self.param_steps else 0

    def zero_grad(self):
        self.flat.zero_grad()

    def _runs(self):
        """contiguous [lo, hi) element ranges of parameters that received a gradient, with their (common) new step
        count: one range covering everything in the usual case"""
        runs = []
        params, offs = self.flat.params, self.flat.offsets
        for i, (p, off) in enumerate(zip(params, offs)):
            if p.grad is None:
                continue
            self.param_steps[i] += 1
            st, hi = self.param_steps[i], off + p.numel()
            if runs and runs[-1][1] == off and runs[-1][2] == st:
                runs[-1][1] = hi
            else:
                runs.append([off, hi, st])
        return runs

    def _prepare(self):
        lane = ME.wgrad_lane(self.flat.grad.device) if self.flat.grad.is_cuda else None
        if lane is not None:   # normally joined already by the engine callback at the end of backward()
            lane.join()
        self.strays = self.flat.gather_strays()
        self.buckets.finish()
        return 1.0 / self.buckets.world

    def state_dict(self):
        sd = {"lr": self.lr, "base_lr": self.base_lr, "param_steps": list(self.param_steps), "kind": type(self).__name__}
        sd.update({k: getattr(self, k) for k in self._state})
        return sd

    def torch_state_dict(self):
        """the state in torch.optim's layout -- `state` {index: {step, per-parameter tensors}} + `param_groups` with the
        parameters numbered in model.parameters() order -- i.e. what Lightning stores as `optimizer_states[0]` and what
        the reference's `trainer.fit(ckpt_path=...)` (train_lidog.py:298-301) hands to torch.optim.Adam / SGD
        .load_state_dict.  Parameters that never received a gradient have no entry, as in torch."""
        state = {}
        for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            if self.param_steps[i] == 0:
                continue
            entry = {k: getattr(self, k)[off:off + p.numel()].view(p.shape).detach().clone() for k in self._state}
            if "exp_avg" in entry:
                entry["step"] = torch.tensor(float(self.param_steps[i]))
            state[i] = entry
        group = dict(self._torch_group(), lr=self.lr, initial_lr=self.base_lr, params=list(range(len(self.flat.params))))
        return {"state": state, "param_groups": [group]}

    def _load_torch_layout(self, sd):
        """inverse of torch_state_dict: a checkpoint written by the reference (torch.optim state through Lightning)"""
        groups = sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(self.flat.params):
            raise ValueError(f"optimizer state for {len(order)} parameters, the model has {len(self.flat.params)}")
        for k in self._state:
            getattr(self, k).zero_()
        self.param_steps = [0] * len(self.flat.params)
        for pos, idx in enumerate(order):
            entry = sd["state"].get(idx)
            if entry is None:
                continue
            p, off = self.flat.params[pos], self.flat.offsets[pos]
            for k in self._state:
                if entry.get(k) is not None:
                    getattr(self, k)[off:off + p.numel()].copy_(entry[k].reshape(-1).to(getattr(self, k).device))
            step = entry.get("step", 1)
            self.param_steps[pos] = int(step.item() if torch.is_tensor(step) else step)
        self.lr = float(groups[0]["lr"])
        self.base_lr = float(groups[0].get("initial_lr", groups[0]["lr"]))
        self.transposed.refresh()

    def load_state_dict(self, sd):
        if "state" in sd and "param_groups" in sd:     # torch.optim layout (a checkpoint of the reference / Lightning)
            return self._load_torch_layout(sd)
        for k in self._state:
            getattr(self, k).copy_(sd[k].to(getattr(self, k).device))
        self.lr, self.base_lr = float(sd["lr"]), float(sd.get("base_lr", sd["lr"]))
        if "param_steps" in sd:
            self.param_steps = list(sd["param_steps"])
        else:   # round-1 checkpoints stored one global step count
            self.param_steps = [int(sd["steps"])] * len(self.flat.params)
        self.transposed.refresh()   # the model's weights were (re)loaded before this call


class FlatAdam(_FlatOptimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) semantics, one fused HIP kernel
    over the flat buffer (csrc/conv2d.hip:k_adam); `grad_scale` folds the 1/world_size of DDP averaging."""
    _state = ("exp_avg", "exp_avg_sq")

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, group=None,
                 bucket_bytes=32 << 20, local=False):
        super().__init__(model, lr, group, bucket_bytes, local)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)

    def _torch_group(self):
        return {"betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None}

    def step(self):
        scale = self._prepare()
        f = self.flat
        for lo, hi, st in self._runs():
            call("lidog_adam_step", ptr(f.flat[lo:hi]), ptr(f.grad[lo:hi]), ptr(self.exp_avg[lo:hi]),
                 ptr(self.exp_avg_sq[lo:hi]), hi - lo, float(self.lr), float(self.betas[0]), float(self.betas[1]),
                 float(self.eps), float(self.weight_decay), st, float(scale))
        self.transposed.refresh()


class FlatSGD(_FlatOptimizer):
    """torch.optim.SGD(lr, momentum, weight_decay, nesterov=True) semantics (trainer_lighting_2d.py:351-355 with
    momentum 0.98, :26), one fused HIP kernel over the flat buffer (csrc/optim.hip:k_sgd)."""
    _state = ("momentum_buffer",)

    def __init__(self, model, lr=1e-3, momentum=0.98, weight_decay=0.0, nesterov=True, group=None,
                 bucket_bytes=32 << 20, local=False):
        super().__init__(model, lr, group, bucket_bytes, local)
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")   # torch's own check
        self.momentum, self.weight_decay, self.nesterov = momentum, weight_decay, nesterov
        self.momentum_buffer = torch.zeros_like(self.flat.flat)

    def _torch_group(self):
        return {"momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay, "nesterov": self.nesterov,
                "maximize": False, "foreach": None, "differentiable": False, "fused": None}

    def step(self):
        scale = self._prepare()
        f = self.flat
        for lo, hi, _ in self._runs():
            call("lidog_sgd_step", ptr(f.flat[lo:hi]), ptr(f.grad[lo:hi]), ptr(self.momentum_buffer[lo:hi]), hi - lo,
                 float(self.lr), float(self.momentum), float(self.weight_decay), 1 if self.nesterov else 0,
                 float(scale))
        self.transposed.refresh()


def make_optimizer(name, model, lr, weight_decay=1e-4, momentum=0.98, group=None):
    """optimizer_name of the reference's pipelines: 'Adam' | 'SGD' (anything else raises, as the reference does)"""
    if name == "Adam":
        return FlatAdam(model, lr=lr, weight_decay=weight_decay, group=group)
    if name == "SGD":
        return FlatSGD(model, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True, group=group)
    raise NotImplementedError(name)


# ------------------------------------------------------------------ learning-rate schedules (stepped per epoch)
class _Scheduler:
    """torch.optim.lr_scheduler semantics for ONE parameter group: construction performs the initial step
    (last_epoch 0), `step()` advances one epoch and writes optimizer.lr.  Same recurrences in python doubles as
    torch's `get_lr`, so the sequence is identical (tests/test_optim_cpu.py compares with torch itself)."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.base_lr = optimizer.base_lr
        self.last_epoch = 0
        optimizer.lr = self._lr(optimizer.lr)

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self._lr(self.optimizer.lr)
        return self.optimizer.lr

    def state_dict(self):
        # last_epoch / base_lrs / _last_lr are the keys of torch's schedulers (what Lightning stores and restores)
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr, "base_lrs": [self.base_lr],
                "_last_lr": [self.optimizer.lr], "kind": type(self).__name__}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self.base_lr = float(sd["base_lr"] if "base_lr" in sd else sd["base_lrs"][0])


class CosineAnnealingLR(_Scheduler):
    def __init__(self, optimizer, T_max=10, eta_min=0.0):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer)

    def _lr(self, lr):
        t, T, lo = self.last_epoch, self.T_max, self.eta_min
        if t == 0:
            return lr
        if (t - 1 - T) % (2 * T) == 0:
            return lr + (self.base_lr - lo) * (1 - math.cos(math.pi / T)) / 2
        return (1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T)) * (lr - lo) + lo


class ExponentialLR(_Scheduler):
    def __init__(self, optimizer, gamma=0.99):
        self.gamma = gamma
        super().__init__(optimizer)

    def _lr(self, lr):
        return lr if self.last_epoch == 0 else lr * self.gamma


class CyclicLR(_Scheduler):
    """mode 'triangular2' (scale 1 / 2^(cycle-1) per cycle), cycle_momentum=False"""

    def __init__(self, optimizer, base_lr, max_lr, step_size_up=5, step_size_down=None):
        up = float(step_size_up)
        down = float(step_size_down) if step_size_down is not None else up
        self.total_size = up + down
        self.step_ratio = up / self.total_size
        self.cyc_base, self.cyc_max = float(base_lr), float(max_lr)
        super().__init__(optimizer)

    def _lr(self, lr):
        cycle = math.floor(1 + self.last_epoch / self.total_size)
        x = 1.0 + self.last_epoch / self.total_size - cycle
        scale = x / self.step_ratio if x <= self.step_ratio else (x - 1) / (self.step_ratio - 1)
        height = (self.cyc_max - self.cyc_base) * scale
        return self.cyc_base + height * (1.0 / (2.0 ** (cycle - 1)))


def make_scheduler(name, optimizer):
    """scheduler_name of the reference's pipelines with the reference's hyper-parameters
    (trainer_lighting_2d.py:379-389); None -> no schedule"""
    if name is None:
        return None
    if name == "CosineAnnealingLR":
        return CosineAnnealingLR(optimizer, T_max=10)
    if name == "ExponentialLR":
        return ExponentialLR(optimizer, gamma=0.99)
    if name == "CyclicLR":
        return CyclicLR(optimizer, base_lr=optimizer.base_lr / 10000, max_lr=optimizer.base_lr, step_size_up=5)
    raise NotImplementedError(name)


def shard_indices(n, rank, world, shuffle=False, seed=0, epoch=0, drop_last=False):
    """torch.utils.data.DistributedSampler semantics (what Lightning injects under strategy='ddp'): optional
    permutation seeded with seed + epoch, padded by wrap-around to a multiple of `world` (so every rank gets the
    same number of samples and no rank runs out of SyncBatchNorm partners), rank-strided."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % world:
        total = (n // world) * world
        idx = idx[:total]
    else:
        total = -(-n // world) * world
        pad = total - len(idx)
        if pad > 0 and idx:
            idx += (idx * -(-pad // len(idx)))[:pad]
    return idx[rank:total:world]
