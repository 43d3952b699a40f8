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


def save_lightning_checkpoint(model, path, epoch=0, global_step=0, optimizer=None):
    """writes the keys eval_target.py / --auto_resume of the reference read back"""
    ckpt = {"epoch": epoch, "global_step": global_step,
            "state_dict": {"model." + k: v.detach().cpu() for k, v in model.state_dict().items()}}
    if optimizer is not None:
        ckpt["optimizer_states"] = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()}]
    torch.save(ckpt, path)
