"""Checkpoint interchange with the reference's Lightning runs.

Lightning's ModelCheckpoint (train_lidog.py:222-225) stores `state_dict` with the LightningModule attribute
prefix `model.` (utils/pipelines/trainer_lighting_2d.py:42); parameter names below that prefix are identical to
lidog_amd.MinkUNet34 / MinkUNet34BEV (checked against the reference classes in tests/golden)."""
import torch


def model_state_dict(ckpt):
    """state_dict of the bare model from a Lightning checkpoint dict (or a plain state_dict)"""
    sd = ckpt.get("state_dict", ckpt)
    if any(k.startswith("model.") for k in sd):
        sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    return sd


def load_lightning_checkpoint(model, path, strict=True, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    missing = model.load_state_dict(model_state_dict(ckpt), strict=strict)
    return ckpt.get("epoch"), missing


def _cpu(v):
    if torch.is_tensor(v):
        return v.detach().cpu()
    if isinstance(v, dict):
        return {k: _cpu(x) for k, x in v.items()}
    return v


def save_lightning_checkpoint(model, path, epoch=0, global_step=0, optimizer=None, scheduler=None):
    """writes the keys eval_target.py / --auto_resume of the reference read back (`epoch`, `global_step`,
    `state_dict` with the `model.` prefix) plus Lightning's `optimizer_states` / `lr_schedulers` lists
    (trainer.fit(ckpt_path=...), train_lidog.py:298-301).  `optimizer_states[0]` is in torch.optim's own layout
    (`state` + `param_groups`, lidog_amd.optim._FlatOptimizer.torch_state_dict), so torch.optim.Adam / SGD over the
    reference model's parameters can load it and a checkpoint written by the reference loads here; the scheduler entry
    carries `last_epoch` (the only field both sides need)."""
    ckpt = {"epoch": epoch, "global_step": global_step,
            "state_dict": {"model." + k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if optimizer is not None:
        to_torch = getattr(optimizer, "torch_state_dict", None)
        ckpt["optimizer_states"] = [_cpu(to_torch() if to_torch is not None else optimizer.state_dict())]
    if scheduler is not None:
        ckpt["lr_schedulers"] = [scheduler.state_dict()]
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    import os
    os.replace(tmp, path)     # a killed run never leaves a truncated checkpoint behind for --auto_resume


def load_training_checkpoint(model, path, optimizer=None, scheduler=None, map_location="cpu"):
    """model + optimiser state + scheduler position of a checkpoint written by save_lightning_checkpoint; returns the
    checkpoint dict (`epoch` = the epoch that had FINISHED when it was written: training resumes at epoch + 1)"""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(model_state_dict(ckpt), strict=True)
    if optimizer is not None and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    if scheduler is not None and ckpt.get("lr_schedulers"):
        scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return ckpt
