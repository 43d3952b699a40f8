"""DICE losses of the LiDOG step, kept on the device (the reference moves logits to the CPU first).

  SoftDICELoss  utils/losses/losses.py:100-109,129-187  (powerize, present-class mask, eps = 0.05)
  DICELoss      utils/losses/losses.py:56-97            (hard one-hot, no mask)
Same arithmetic order as the reference; only the `.cpu()` round trips (losses.py:72-73,148-149) are gone.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _dice(prob, target_w, present, powerize):
    inter = (prob * target_w).sum(dim=0)
    union = ((prob.pow(2) if powerize else prob).sum(dim=0) + target_w.sum(dim=0)) + 1e-12
    iou = (present * 2 * inter / union).sum(dim=0) / (present.sum(dim=0) + 1e-12)
    return 1 - iou.mean()


class SoftDICELoss(nn.Module):
    def __init__(self, ignore_label=None, powerize=True, use_tmask=True, neg_range=False, eps=0.05, is_kitti=False):
        super().__init__()
        if is_kitti:
            raise NotImplementedError("the 19-class KITTI soft-label variant is not on the LiDOG hot path")
        self.ignore_label, self.powerize, self.use_tmask, self.neg_range, self.eps = \
            ignore_label, powerize, use_tmask, neg_range, eps

    def forward(self, output, target):
        if self.ignore_label is not None:
            valid = target != self.ignore_label
            target, output = target[valid], output[valid, :]
        C = output.shape[1]
        onehot = F.one_hot(target, num_classes=C)
        soft = torch.where(onehot == 1, 1 - self.eps, self.eps / (C - 1)).to(torch.float32)
        prob = F.softmax(output, dim=-1)
        present = (onehot.sum(dim=0) > 0).int() if self.use_tmask else torch.ones(C, dtype=torch.int32,
                                                                                  device=output.device)
        loss = _dice(prob, soft, present, self.powerize)
        return loss - 1 if self.neg_range else loss


class DICELoss(nn.Module):
    def __init__(self, ignore_label=None, powerize=False, use_tmask=False):
        super().__init__()
        self.ignore_label, self.powerize, self.use_tmask = ignore_label, powerize, use_tmask

    def forward(self, output, target):
        if self.ignore_label is not None:
            valid = target != self.ignore_label
            target, output = target[valid], output[valid, :]
        C = output.shape[1]
        onehot = F.one_hot(target, num_classes=C)
        prob = F.softmax(output, dim=-1)
        present = (onehot.sum(dim=0) > 0).int() if self.use_tmask else torch.ones(C, dtype=torch.int32,
                                                                                  device=output.device)
        return _dice(prob, onehot, present, self.powerize)
