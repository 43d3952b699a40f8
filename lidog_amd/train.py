"""The owning training driver: model -> data-parallel wiring -> epochs of steps -> validation -> checkpoints.

What the reference's entry scripts do through pytorch-lightning, as plain arguments (no YAML engine):
  get_model / SyncBN + DDP / Trainer args   train_lidog.py:42-75,227-231,286-296 (train_source.py:43-58,196-234,
                                            train_aug_based.py:44-53,189-230)
  ModelCheckpoint(every_n_epochs=1, save_on_train_epoch_end, save_top_k=-1)          train_lidog.py:222-225
  Trainer(max_epochs, check_val_every_n_epoch, num_sanity_val_steps=2), fit(ckpt_path=resume)   :286-301
  DistributedSampler sharding + per-epoch reshuffle (what Lightning injects under strategy='ddp', shuffle=True :186)
  epoch-interval scheduler stepping (configure_optimizers returns ([optimizer], [scheduler]))

    python -m lidog_amd.train --model MinkUNet34BEV --epochs 2 --scans 16 --batch 4 --save-dir /tmp/run
    python -m torch.distributed.run --nproc-per-node N -m lidog_amd.train ...        (one process per GPU, RCCL)

Scans are synthetic (lidog_amd.synth; there are no datasets on the box); anything with `__len__` and
`batch(indices, device) -> dict` (keys of CollateFNSingleSourceBEVMultiLevel, collation.py:318-325) can be passed as
`train_data` / `val_data` instead.
"""
import argparse
import os
import re

import torch
import torch.distributed as dist

from . import me as ME
from . import synth
from .checkpoint import load_training_checkpoint, save_lightning_checkpoint
from .evaluate import per_class_iou
from .optim import make_optimizer, make_scheduler, shard_indices
from .trainer import LiDOGStep, SourceStep, setup_data_parallel


class SynthScans:
    """`n` synthetic scans of one configuration; scan i = seed `first + i` (SURVEY.md 8(d) generator)"""

    def __init__(self, n, config="kitti120k", first=0, mix3d=False, bev_size=167):
        self.n, self.config, self.first, self.mix3d, self.bev_size = n, config, first, mix3d, bev_size

    def __len__(self):
        return self.n

    def batch(self, indices, device):
        return synth.make_batch([self.first + i for i in indices], self.config, device, bev_size=self.bev_size,
                                mix3d=self.mix3d)


def bev_image_size(bound_2d, voxel=0.05, pool=(5, 3, 1)):
    """side of the BEV logits: sparse2super's H = int(2B / voxel) (minkunet_bev.py:184-185) through MaxPool2d(5, 3, 1)
    and the two stride-2 convolutions of Encoder2D: 167 for B = 50 (bev_img_sizes, semantickitti.yaml:8), 100 for 30"""
    h = int(2 * bound_2d / voxel)
    h = (h + 2 * pool[2] - pool[0]) // pool[1] + 1
    for _ in range(2):
        h = (h + 2 - 3) // 2 + 1
    return h


def build_model(kind="MinkUNet34BEV", bound_2d=50.0, in_channels=1, out_channels=7, conv1_kernel_size=5,
                decoder_2d_levels=("block8",), device="cuda"):
    """get_model of train_lidog.py:42-75 (MinkUNet34BEV) / train_source.py:43-58 (MinkUNet34)"""
    import lidog_amd
    if kind == "MinkUNet34BEV":
        m = lidog_amd.MinkUNet34BEV(in_channels=in_channels, out_channels=out_channels, D=3,
                                    initial_kernel_size=conv1_kernel_size, decoder_2d_level=list(decoder_2d_levels),
                                    mapping_bound_2d=bound_2d)
    elif kind == "MinkUNet34":
        m = lidog_amd.MinkUNet34(in_channels=in_channels, out_channels=out_channels, D=3,
                                 initial_kernel_size=conv1_kernel_size)
    else:
        raise NotImplementedError(kind)
    return m.to(device)


def build_step(model, kind="MinkUNet34BEV", optimizer="Adam", lr=1e-3, scheduler=None, weight_decay=1e-4,
               momentum=0.98, warmup_epochs=0, source_weights=(0.5, 0.5), num_classes=7, ignore_label=-1):
    """SyncBN conversion when data-parallel (train_lidog.py:227-231), optimiser + scheduler
    (trainer_lighting_2d.py:349-394), step object (PLTTrainer2D / PLTTrainer).  Returns (model, step, scheduler)."""
    model = setup_data_parallel(model)
    model.train()
    opt = make_optimizer(optimizer, model, lr, weight_decay=weight_decay, momentum=momentum)
    sched = make_scheduler(scheduler, opt)
    if kind == "MinkUNet34BEV":
        step = LiDOGStep(model, opt, source_weights=source_weights, warmup_epochs=warmup_epochs,
                         num_classes=num_classes, ignore_label=ignore_label)
    else:
        step = SourceStep(model, opt, ignore_label=ignore_label)
    return model, step, sched


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def last_checkpoint(save_dir):
    """--auto_resume of train_lidog.py:142-172: the checkpoint with the highest epoch under save_dir/checkpoints"""
    d = os.path.join(save_dir, "checkpoints")
    best = None
    if os.path.isdir(d):
        for f in os.listdir(d):
            m = re.match(r"epoch=(\d+)-step=(\d+)\.ckpt$", f)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), os.path.join(d, f))
    return best[1] if best else None


class Fit:
    """trainer.fit(pl_module, train_dataloaders, val_dataloaders, ckpt_path) of train_lidog.py:286-301."""

    def __init__(self, model_kind="MinkUNet34BEV", bound_2d=50.0, batch_size=4, optimizer="Adam", lr=1e-3,
                 scheduler=None, epochs=25, warmup_epochs=0, source_weights=(0.5, 0.5), weight_decay=1e-4,
                 momentum=0.98, check_val_every_n_epoch=5, num_sanity_val_steps=2, save_dir=None, seed=1234,
                 train_data=None, val_data=None, shuffle=True, resume=None, auto_resume=False, prefetch=True,
                 device="cuda", log=None, state_dict=None):
        self.rank, self.world = _rank_world()
        self.kind, self.batch_size, self.epochs = model_kind, batch_size, epochs
        self.check_val, self.sanity = check_val_every_n_epoch, num_sanity_val_steps
        self.save_dir, self.seed, self.shuffle, self.prefetch = save_dir, seed, shuffle, prefetch
        self.train_data = train_data if train_data is not None else SynthScans(16)
        self.val_data = val_data
        self.device = device
        self.log = log if log is not None else (print if self.rank == 0 else (lambda *_: None))
        torch.manual_seed(seed)                         # pipeline.seed (semantickitti.yaml:32)
        model = build_model(model_kind, bound_2d, device=device)
        if state_dict is not None:
            model.load_state_dict(state_dict)
        self.model, self.step, self.sched = build_step(
            model, model_kind, optimizer, lr, scheduler, weight_decay, momentum, warmup_epochs, source_weights)
        self.opt = self.step.opt
        self.epoch, self.global_step = 0, 0
        self.history = []
        if auto_resume and save_dir and resume is None:
            resume = last_checkpoint(save_dir)
        if resume:
            ck = load_training_checkpoint(self.model, resume, self.opt, self.sched, map_location=device)
            self.epoch, self.global_step = ck["epoch"] + 1, ck["global_step"]
            self.log(f"resumed from {resume}: next epoch {self.epoch}, global step {self.global_step}")

    # ------------------------------------------------------------------ data
    def _epoch_batches(self, data, epoch, shuffle):
        idx = shard_indices(len(data), self.rank, self.world, shuffle=shuffle, seed=self.seed, epoch=epoch)
        return [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]

    # ------------------------------------------------------------------ validation (validation_step, :295-328)
    @torch.no_grad()
    def validation_step(self, batch):
        was = self.model.training
        self.model.eval()
        st = ME.SparseTensor(coordinates=batch["coords_int"], features=batch["source_features0"])
        out = self.model(st)
        logits = (out[0] if isinstance(out, tuple) else out).F
        labels = batch["source_sem_labels0"].long()
        crit = self.step.sem_criterion if hasattr(self.step, "sem_criterion") else self.step.criterion
        loss = crit(logits, labels)
        iou = per_class_iou(logits.max(dim=1)[1], labels)      # sklearn jaccard_score(labels=0..C-1), -1 = absent
        self.model.train(was)
        present = iou >= 0
        return {"sem_loss": float(loss), "source_iou": float(iou[present].mean()) if bool(present.any()) else 0.0,
                "per_class_iou": iou.tolist()}

    def validate(self, epoch, limit=None):
        if self.val_data is None:
            return None
        res = []
        for ids in self._epoch_batches(self.val_data, 0, False)[:limit]:
            res.append(self.validation_step(self.val_data.batch(ids, self.device)))
        if not res:
            return None
        out = {"epoch": epoch, "sem_loss": sum(r["sem_loss"] for r in res) / len(res),
               "source_iou": sum(r["source_iou"] for r in res) / len(res), "steps": len(res)}
        if self.world > 1:   # sync_dist=True of log_losses (trainer_lighting_2d.py:330-347): mean over ranks
            t = torch.tensor([out["sem_loss"], out["source_iou"]], device=self.device, dtype=torch.float64)
            dist.all_reduce(t)
            out["sem_loss"], out["source_iou"] = (t / self.world).tolist()
        return out

    # ------------------------------------------------------------------ checkpoints (ModelCheckpoint, :222-225)
    def save(self, epoch):
        if self.save_dir is None or self.rank != 0:
            return None
        d = os.path.join(self.save_dir, "checkpoints")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"epoch={epoch}-step={self.global_step}.ckpt")
        save_lightning_checkpoint(self.model, path, epoch=epoch, global_step=self.global_step, optimizer=self.opt,
                                  scheduler=self.sched)
        return path

    # ------------------------------------------------------------------ the loop
    def run(self):
        if self.val_data is not None and self.sanity > 0 and self.epoch == 0:
            s = self.validate(-1, limit=self.sanity)            # num_sanity_val_steps=2 (train_lidog.py:294)
            self.log(f"sanity validation: {s}")
        for epoch in range(self.epoch, self.epochs):
            batches = self._epoch_batches(self.train_data, epoch, self.shuffle)
            cur = self.train_data.batch(batches[0], self.device) if batches else None
            losses = []
            for i in range(len(batches)):
                nxt = self.train_data.batch(batches[i + 1], self.device) if i + 1 < len(batches) else None
                out = self.step.training_step(cur, epoch=epoch, prefetch=nxt if self.prefetch else None)
                losses.append(out["loss"])
                self.global_step += 1
                cur = nxt
            lr_used = self.opt.lr
            if self.sched is not None:
                self.sched.step()                               # interval='epoch'
            rec = {"epoch": epoch, "global_step": self.global_step, "lr": lr_used,
                   "loss": float(torch.stack(losses).mean()) if losses else float("nan"),
                   "losses": [float(l) for l in losses]}
            if self.val_data is not None and (epoch + 1) % self.check_val == 0:
                rec["validation"] = self.validate(epoch)
            if dist.is_available() and dist.is_initialized():
                from .comm import _TRANSPORTS
                for tr in _TRANSPORTS.values():                 # a statistics message that never arrived invalidates the epoch
                    tr.check()
            rec["checkpoint"] = self.save(epoch)                # every_n_epochs=1, save_top_k=-1
            self.history.append(rec)
            self.log({k: v for k, v in rec.items() if k != "losses"})
            self.epoch = epoch + 1
        if self.world > 1:
            dist.barrier()
        return self.history


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", default="MinkUNet34BEV", choices=["MinkUNet34BEV", "MinkUNet34"])
    ap.add_argument("--bound", type=float, default=50.0)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--optimizer", default="Adam", choices=["Adam", "SGD"])
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--scheduler", default=None, choices=[None, "CosineAnnealingLR", "ExponentialLR", "CyclicLR"])
    ap.add_argument("--epochs", type=int, default=25)
    ap.add_argument("--warmup-epochs", type=int, default=0)
    ap.add_argument("--scans", type=int, default=16, help="synthetic training scans per epoch (all ranks together)")
    ap.add_argument("--val-scans", type=int, default=0)
    ap.add_argument("--config", default="kitti120k", choices=sorted(synth.CONFIGS))
    ap.add_argument("--mix3d", action="store_true")
    ap.add_argument("--check-val-every-n-epoch", type=int, default=5)
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--auto-resume", action="store_true")
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    bev = bev_image_size(a.bound)
    fit = Fit(a.model, a.bound, a.batch, a.optimizer, a.lr, a.scheduler, a.epochs, a.warmup_epochs,
              check_val_every_n_epoch=a.check_val_every_n_epoch, save_dir=a.save_dir, seed=a.seed,
              train_data=SynthScans(a.scans, a.config, mix3d=a.mix3d, bev_size=bev),
              val_data=SynthScans(a.val_scans, a.config, first=10 ** 6, mix3d=a.mix3d, bev_size=bev) if a.val_scans else None,
              resume=a.resume, auto_resume=a.auto_resume)
    fit.run()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        from . import comm
        comm.reset()                 # this library's RCCL communicators / mailboxes go before torch's group does
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
