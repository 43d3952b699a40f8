"""Torch-CPU restatements of the reference's OWN python on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/me_oracle.c header).  These functions
restate reference code that cannot travel to the GPU box; each one is checked
against golden vectors the reference function itself produced in the build
container (tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_cpu.py).

  sparse2super_ref   utils/models/minkunet_bev.py:158-230 (filter_bounds + sparse2super)
  Encoder2DRef       utils/models/conv2d.py:9-25,42-52,113-119,180-197
  soft_dice_loss_ref utils/losses/losses.py:100-109,129-187
  dice_loss_ref      utils/losses/losses.py:56-97
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def bev_image_size(bound, voxel=0.05):
    """max_height / max_width of sparse2super (minkunet_bev.py:184-185)."""
    return int(torch.tensor((bound - (-bound)) / voxel).int())


def bev_pixels_ref(coords_xyz, bound, voxel=0.05):
    """float32 index math of minkunet_bev.py:176,163-165,213-214 for int32 voxel coords [n,3].
    Returns (in_bounds bool [n], pixel_x long [n], pixel_y long [n]) -- pixels valid where in_bounds."""
    xyz = coords_xyz * voxel  # int32 tensor * python float -> float32
    lo, hi = -bound, bound
    inb = torch.logical_and(torch.logical_and(lo < xyz[:, 0], xyz[:, 0] < hi),
                            torch.logical_and(lo < xyz[:, 1], xyz[:, 1] < hi))
    H = torch.tensor((hi - lo) / voxel).int()
    px = torch.floor((xyz[:, 0] - lo) / voxel).long()
    py = torch.floor(H - (xyz[:, 1] - lo) / voxel).long() - 1
    # a negative index wraps in `dense[py, px] = feats` (float32 rounding gives py = -1 just inside the bound)
    py = torch.where(py < 0, py + int(H), py)
    px = torch.where(px < 0, px + int(H), px)
    return inb, px, py


def sparse2super_ref(C, Fe, bound, voxel=0.05, pool=(5, 3, 1)):
    """C int32 [N,4] (b,x,y,z), Fe float32 [N,Cf] (may require grad) -> [B,Cf,Ho,Wo].

    Duplicate (py,px) targets: the LAST row wins (sequential index_put_, the
    deterministic behaviour of minkunet_bev.py:217 on one CPU thread)."""
    C = C.cpu()
    B = int(C[:, 0].max()) + 1
    H = bev_image_size(bound, voxel)
    Cf = Fe.shape[-1]
    outs = []
    for b in range(B):
        sel = C[:, 0] == b
        fb = Fe[sel]
        inb, px, py = bev_pixels_ref(C[sel][:, 1:], bound, voxel)
        fb, px, py = fb[inb], px[inb], py[inb]
        # last-write-wins made explicit so the result does not depend on torch's index_put_ kernel
        lin = py * H + px
        order = torch.arange(lin.shape[0])
        winner = torch.full((H * H,), -1, dtype=torch.long)
        winner.scatter_reduce_(0, lin, order, reduce="amax", include_self=True)
        keep = winner[lin] == order
        dense = torch.zeros((H * H, Cf), dtype=Fe.dtype)
        # every duplicate row receives the pixel's gradient in the reference (index_put backward is a
        # gather); emulate with a custom function below
        dense = _PutLastWins.apply(dense, lin, fb, keep)
        dense = dense.view(1, -1, H, H)  # [H,W,C] memory reinterpreted as [1,C,H,W] (minkunet_bev.py:221)
        outs.append(F.max_pool2d(dense, *pool))
    return torch.cat(outs, dim=0)


class _PutLastWins(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, lin, feats, keep):
        ctx.save_for_backward(lin)
        out = dense.clone()
        out[lin[keep]] = feats[keep]
        return out

    @staticmethod
    def backward(ctx, g):
        (lin,) = ctx.saved_tensors
        return None, None, g[lin], None


class Encoder2DRef(nn.Module):
    """Same modules and state_dict keys as the reference's Encoder2D(input_size, n_classes)."""

    def __init__(self, input_size, n_classes=7, binary_seg=False):
        super().__init__()
        assert not binary_seg, "binary_seg is off in every LiDOG config"
        dc = nn.Sequential(
            nn.Conv2d(input_size, 256, kernel_size=3, padding=1, stride=2, bias=False),
            nn.BatchNorm2d(256),
            nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, kernel_size=3, padding=1, stride=2, bias=False),
            nn.BatchNorm2d(256),
            nn.ReLU(inplace=True))
        dbl = nn.Module()
        dbl.double_conv = dc
        down = nn.Module()
        down.maxpool_conv = nn.Sequential(dbl)
        self.down1 = down
        oc = nn.Module()
        oc.conv = nn.Conv2d(256, n_classes, kernel_size=1)
        self.out_conv = oc

    def forward(self, x):
        x = self.down1.maxpool_conv[0].double_conv(x)
        return self.out_conv.conv(x)


def _dice(output, target_w, target_onehot, powerize, use_tmask):
    output = F.softmax(output, dim=-1)
    inter = (output * target_w).sum(dim=0)
    if powerize:
        union = (output.pow(2).sum(dim=0) + target_w.sum(dim=0)) + 1e-12
    else:
        union = (output.sum(dim=0) + target_w.sum(dim=0)) + 1e-12
    if use_tmask:
        tmask = (target_onehot.sum(dim=0) > 0).int()
    else:
        tmask = torch.ones(target_onehot.shape[1]).int()
    iou = (tmask * 2 * inter / union).sum(dim=0) / (tmask.sum(dim=0) + 1e-12)
    return 1 - iou.mean()


def soft_dice_loss_ref(output, target, ignore_label=-1, eps=0.05):
    """SoftDICELoss(ignore_label) defaults: powerize=True, use_tmask=True, eps=0.05."""
    output, target = output.cpu(), target.cpu()
    valid = torch.logical_not(target == ignore_label)
    target, output = target[valid], output[valid, :]
    onehot = F.one_hot(target, num_classes=output.shape[1])
    soft = torch.empty(onehot.shape)
    soft[onehot == 0] = eps / (onehot.shape[-1] - 1)
    soft[onehot == 1] = 1 - eps
    return _dice(output, soft, onehot, True, True)


def dice_loss_ref(output, target, ignore_label=-1):
    """DICELoss(ignore_label) defaults: powerize=False, use_tmask=False."""
    output, target = output.cpu(), target.cpu()
    valid = torch.logical_not(target == ignore_label)
    target, output = target[valid], output[valid, :]
    onehot = F.one_hot(target, num_classes=output.shape[1])
    return _dice(output, onehot, onehot, False, False)
