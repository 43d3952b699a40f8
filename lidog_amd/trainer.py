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
import os as _os

import torch
import torch.distributed as dist

from . import me as ME
from .losses import DICELoss, SoftDICELoss
from .optim import (FlatAdam, FlatParams, FlatSGD, GradientBuckets, make_optimizer, make_scheduler,  # noqa: F401
                    shard_indices)


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


_PEER_CHECK_EVERY = int(_os.environ.get("LIDOG_PEER_CHECK_EVERY", "200"))


def _check_transport(step):
    """every LIDOG_PEER_CHECK_EVERY steps of a data-parallel run whose statistics take the peer all-reduce: its error
    word, agreed over the ranks (comm.Transport.check: collective, synchronises with the device) -- a rank that stopped
    sending is reported by every rank within that many steps instead of at the next epoch boundary"""
    step._n_steps = getattr(step, "_n_steps", 0) + 1
    if step._n_steps % _PEER_CHECK_EVERY or not (dist.is_available() and dist.is_initialized()):
        return
    from .comm import _TRANSPORTS
    for tr in _TRANSPORTS.values():
        if tr.peer is not None:
            tr.check()


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
        total, sem_loss, bev_loss, sem = self.forward_loss(batch, epoch)
        self.last_path = type(sem.F.grad_fn).__name__     # "_TrunkFnBackward": the trunk executor took the pass
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        self._after_step(prefetch, prefetch_ready)
        _check_transport(self)
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
        _check_transport(self)
        return {"loss": loss.detach()}


def setup_data_parallel(model):
    """train_lidog.py:227-231: SyncBatchNorm conversion of the sparse BNs when world_size > 1 (the two
    BatchNorm2d of Encoder2D stay per-rank, as in the reference)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or ME.MinkowskiSyncBatchNorm.single_rank):
        model = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    return model
