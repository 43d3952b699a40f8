"""Training step of LiDOG on the GPU: loss composition, Adam on a flat HBM buffer, RCCL data parallelism.

Restates what the reference gets from pytorch-lightning (not installable here, and python that the GPU
box never receives):
  training_step        utils/pipelines/trainer_lighting_2d.py:141-201  (PLTTrainer2D)
                       utils/pipelines/trainer_lighting.py:92-104      (PLTTrainer, source-only)
  configure_optimizers utils/pipelines/trainer_lighting_2d.py:349-394  (Adam lr, weight_decay 1e-4)
  DDP + SyncBN         train_lidog.py:227-231,286-289                   (strategy='ddp')

One process per GPU; gradients live in ONE contiguous fp32 buffer that is all-reduced over RCCL in
a few large buckets launched as soon as their last gradient is produced (overlapping the rest of
backward), then consumed by a single fused Adam kernel.
"""
import torch
import torch.distributed as dist

from . import me as ME
from ._lib import call, ptr
from .losses import DICELoss, SoftDICELoss


class FlatParams:
    """Re-homes every parameter (and its .grad) of `model` into two contiguous fp32 buffers."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            p._flat_ref = (self.grad, off)   # lets backward kernels write this gradient in place (me._grad_out)
            self.offsets.append(off)
            off += n
        self.total = total

    def zero_grad(self):
        """set_to_none semantics: backward kernels write fresh views of the flat buffer and autograd adopts them
        as .grad, so there is no `grad += new` pass; the buffer is cleared for parameters that get no gradient"""
        self.grad.zero_()
        for p in self.params:
            p.grad = None

    def in_place(self, p, off):
        return p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * off

    def gather_strays(self):
        """gradients that autograd produced outside the flat buffer (e.g. a bias gradient from a torch op) are
        copied in with one fused call; returns how many there were"""
        dst, src = [], []
        for p, off in zip(self.params, self.offsets):
            if p.grad is not None and not self.in_place(p, off):
                view = self.grad[off:off + p.numel()].view(p.shape)
                dst.append(view)
                src.append(p.grad)
                p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)
        return len(dst)


class GradientBuckets:
    """Bucketed all-reduce (sum) of the flat gradient buffer, overlapped with backward.

    Parameters are registered in forward order, gradients arrive roughly in reverse, so buckets are
    contiguous slices walked from the END of the buffer.  xGMI is point-to-point (7 links per GPU):
    a few large messages (default 32 MiB) keep every link busy without paying per-message latency."""

    single_rank = False   # test hook: bucket and all-reduce even in a one-rank process group

    def __init__(self, flat, group=None, bucket_bytes=32 << 20):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (self.single_rank and dist.is_initialized())
        self.handles = []
        self.deferred = []
        self.index_of = {id(p): i for i, p in enumerate(flat.params)}
        self.bucket_of = {}
        self.pending0 = []
        self.slices = []
        if not self.active:
            return
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
        self.pending = list(self.pending0)
        for p in flat.params:
            p.register_post_accumulate_grad_hook(self._hook)

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
            self._reduce(b)

    def _reduce(self, b):
        lo, hi = self.slices[b]
        buf = self.flat.grad[lo:hi]
        if buf.is_cuda and ME._WgradLane.active():
            # Weight gradients of this bucket may still be running on the second stream of the backward pass, and
            # a collective queued now would sit in RCCL's stream in front of every later SyncBatchNorm all-reduce
            # until they are done, stalling the data-gradient chain behind the weight gradients.  The bucket is
            # reduced after the join instead (finish()); 155 MB over xGMI is ~1 ms at 8 GPUs.
            self.deferred.append(b)
            return
        self.handles.append(dist.all_reduce(buf, group=self.group, async_op=True))

    def finish(self):
        """wait for every bucket; buckets whose hooks did not all fire (unused parameters) are reduced now"""
        if not self.active:
            return
        for b, left in enumerate(self.pending):
            if left > 0:
                self._reduce(b)
        for b in self.deferred:   # the engine callback of the second stream has joined it by now
            lo, hi = self.slices[b]
            self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], group=self.group, async_op=True))
        self.deferred = []
        for h in self.handles:
            h.wait()
        self.handles = []
        self.pending = list(self.pending0)


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) semantics, one fused HIP kernel
    over the flat buffer (csrc/conv2d.hip:k_adam); `grad_scale` folds the 1/world_size of DDP averaging."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, group=None,
                 bucket_bytes=32 << 20):
        self.flat = FlatParams(model)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat)
        self.steps = 0
        self.buckets = GradientBuckets(self.flat, group, bucket_bytes)

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        lane = ME.wgrad_lane(self.flat.grad.device) if self.flat.grad.is_cuda else None
        if lane is not None:   # normally joined already by the engine callback at the end of backward()
            lane.join()
        self.strays = self.flat.gather_strays()
        self.buckets.finish()
        self.steps += 1
        scale = 1.0 / self.buckets.world
        call("lidog_adam_step", ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             self.flat.total, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             float(self.weight_decay), self.steps, float(scale))

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "steps": self.steps, "lr": self.lr}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.steps, self.lr = sd["steps"], sd["lr"]


class _CoordinatePrefetch:
    """Coordinate maps of the NEXT batch are built on a side stream while the GPU still works on the current step
    (the host runs ahead of the GPU by then): `training_step(batch, prefetch=next_batch)`.  What to build is the
    trace of map uses recorded by the first forward pass; it is the data-loader-side half of the reference's step
    (ME builds its maps inside the forward call) moved off the critical path, not skipped."""

    _trace = None

    @staticmethod
    def _coords(batch):
        return batch["coords_int"] if "coords_int" in batch else batch["source_coordinates0"].int()

    def _sparse_input(self, batch):
        coords = self._coords(batch)
        hit = self.__dict__.setdefault("_prepared", {}).pop(id(coords), None)
        if hit is not None and hit[0] is coords:
            st = ME.SparseTensor(features=batch["source_features0"], coordinates=coords, coordinate_manager=hit[1])
        else:
            st = ME.SparseTensor(coordinates=coords, features=batch["source_features0"])
        self._last_manager = st.coordinate_manager
        return st

    def _after_step(self, prefetch, prefetch_ready):
        self._trace = self._last_manager.trace
        if prefetch is not None and "coords_int" in prefetch:
            coords = prefetch["coords_int"]
            self.__dict__.setdefault("_prepared", {})[id(coords)] = \
                (coords, ME.CoordinateManager.prepare(coords, self._trace, prefetch_ready))


class LiDOGStep(_CoordinatePrefetch):
    """PLTTrainer2D.training_step without the host round trips (coords / logits stay in HBM)."""

    def __init__(self, model, optimizer, source_weights=(0.5, 0.5), warmup_epochs=0, num_classes=7, ignore_label=-1):
        self.model, self.opt = model, optimizer
        self.w, self.warmup, self.nc = source_weights, warmup_epochs, num_classes
        self.sem_criterion = SoftDICELoss(ignore_label=ignore_label)
        self.bev_criterion = DICELoss(ignore_label=ignore_label)

    def forward_loss(self, batch, epoch=0):
        st = self._sparse_input(batch)
        sem, bev = self.model(st, is_train=True)
        bev_loss = 0.0
        for key, lab in batch["source_bev_labels0"].items():
            # NCHW logits through .view(-1, C): reproduces trainer_lighting_2d.py:181-182 literally
            bev_loss = bev_loss + self.bev_criterion(bev[key].view(-1, self.nc), lab.view(-1)) / len(bev)
        if epoch >= self.warmup:
            sem_loss = self.sem_criterion(sem.F, batch["source_sem_labels0"].long())
            total = self.w[0] * sem_loss + self.w[1] * bev_loss
        else:
            sem_loss = torch.zeros((), device=sem.F.device)
            total = bev_loss
        return total, sem_loss, bev_loss, sem

    def training_step(self, batch, epoch=0, prefetch=None, prefetch_ready=None):
        """`prefetch`: the batch of the NEXT call (its coordinate maps are built while this step still runs on
        the GPU); `prefetch_ready`: event after which its coordinates are valid (None: everything queued so far)"""
        total, sem_loss, bev_loss, _ = self.forward_loss(batch, epoch)
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        self._after_step(prefetch, prefetch_ready)
        return {"loss": total.detach(), "sem_loss": sem_loss.detach(), "bev_loss": bev_loss.detach()}


class SourceStep(_CoordinatePrefetch):
    """PLTTrainer.training_step (train_source.py / Mix3D): MinkUNet34, SoftDICE only."""

    def __init__(self, model, optimizer, ignore_label=-1):
        self.model, self.opt = model, optimizer
        self.criterion = SoftDICELoss(ignore_label=ignore_label)

    def training_step(self, batch, epoch=0, prefetch=None, prefetch_ready=None):
        st = self._sparse_input(batch)
        out = self.model(st, is_seg=True)
        loss = self.criterion(out.F, batch["source_sem_labels0"].long())
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        self._after_step(prefetch, prefetch_ready)
        return {"loss": loss.detach()}


def setup_data_parallel(model):
    """train_lidog.py:227-231: SyncBatchNorm conversion of the sparse BNs when world_size > 1 (the two
    BatchNorm2d of Encoder2D stay per-rank, as in the reference)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or ME.MinkowskiSyncBatchNorm.single_rank):
        model = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    return model


def shard_indices(n, rank, world):
    """DistributedSampler-style rank-strided indices (what Lightning injects under strategy='ddp')"""
    return list(range(rank, n - n % world if n >= world else n, world))
