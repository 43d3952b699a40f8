"""DICE losses of the LiDOG step, kept on the device (the reference moves logits to the CPU first).

  SoftDICELoss  utils/losses/losses.py:100-109,129-187  (powerize, present-class mask, eps = 0.05)
  DICELoss      utils/losses/losses.py:56-97            (hard one-hot, no mask)
Same formulas as the reference, as HIP kernels (csrc/losses.hip; there is no torch formula behind them: CPU tensors and
unsupported class counts raise); the `.cpu()` round trips (losses.py:72-73,148-149) are gone, and rows carrying the
ignore label are skipped in place instead of being compacted away with boolean indexing: the compaction needs the
number of valid rows on the host, i.e. a device synchronisation in the middle of every step.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr

_FUSED_CLASSES = tuple(range(2, 21))   # csrc/losses.hip: one instantiation per class count (the reference uses 7)


class _DiceFn(torch.autograd.Function):
    """csrc/losses.hip: softmax + the per-class sums in one pass, the gradient in one pass"""

    @staticmethod
    def forward(ctx, logits, target, ignore_label, eps, soft, powerize, use_tmask, offset):
        logits, target = logits.contiguous(), target.contiguous()
        if target.dtype != torch.int64:
            target = target.long()
        n, C = logits.shape
        dev = logits.device
        ws = torch.empty(_lib.load().lidog_dice_ws(C), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(2 * C, dtype=torch.float32, device=dev)
        has_ignore = ignore_label is not None
        cfg = (n, C, int(ignore_label) if has_ignore else 0, 1 if has_ignore else 0, float(eps), 1 if soft else 0,
               1 if powerize else 0)
        call("lidog_dice_fwd", ptr(logits), ptr(target), *cfg, 1 if use_tmask else 0, float(offset), ptr(ws),
             ptr(loss), ptr(coef))
        ctx.save_for_backward(logits, target, coef)
        ctx.cfg = cfg
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, target, coef = ctx.saved_tensors
        g = torch.empty_like(logits)
        call("lidog_dice_bwd", ptr(logits), ptr(target), *ctx.cfg, ptr(coef), ptr(gout.contiguous()), ptr(g))
        return g, None, None, None, None, None, None, None


def _check(output):
    """the DICE losses exist as HIP kernels only (csrc/losses.hip): no torch formula behind them"""
    _lib.require_gpu(output, "logits")
    if output.dim() != 2 or output.shape[1] not in _FUSED_CLASSES or output.dtype != torch.float32:
        raise NotImplementedError(f"lidog_amd DICE losses take float32 logits [n, C] with C in {_FUSED_CLASSES}, "
                                  f"got {tuple(output.shape)} {output.dtype}")


class SoftDICELoss(nn.Module):
    def __init__(self, ignore_label=None, powerize=True, use_tmask=True, neg_range=False, eps=0.05, is_kitti=False):
        super().__init__()
        if is_kitti:
            raise NotImplementedError("the 19-class KITTI soft-label variant is not on the LiDOG hot path")
        self.ignore_label, self.powerize, self.use_tmask, self.neg_range, self.eps = \
            ignore_label, powerize, use_tmask, neg_range, eps

    def forward(self, output, target):
        _check(output)
        return _DiceFn.apply(output, target, self.ignore_label, self.eps, True, self.powerize, self.use_tmask,
                             -1.0 if self.neg_range else 0.0)


class DICELoss(nn.Module):
    def __init__(self, ignore_label=None, powerize=False, use_tmask=False):
        super().__init__()
        self.ignore_label, self.powerize, self.use_tmask = ignore_label, powerize, use_tmask

    def forward(self, output, target):
        _check(output)
        return _DiceFn.apply(output, target, self.ignore_label, 0.0, False, self.powerize, self.use_tmask, 0.0)
