"""lidog_amd -- MI355X-native engine for LiDOG's hot path (MinkUNet34 + BEV head).

Public surface:
  lidog_amd.me            MinkowskiEngine-compatible operator API (HIP kernels via the C ABI)
  lidog_amd.bev           sparse2super + Encoder2D (fused BEV projection, MFMA 2-D head)
  lidog_amd.MinkUNet34 / MinkUNet34BEV    the reference models wired to those operators
  lidog_amd.losses        SoftDICELoss / DICELoss on the device
  lidog_amd.trainer       training step, Adam, RCCL data parallelism
  lidog_amd.trunk         the whole encoder-decoder as one launch sequence per pass (csrc/trunk.hip)
"""
import os as _os

# compute, weight-gradient, coordinate-map and gradient-bucket streams (plus RCCL's own for up to three communicators)
# must not share a hardware queue: the runtime's default is 4 queues for all streams of a process, and at 8 the
# data-parallel step still serialised (67.7 vs 50.7 ms per step at 16); only effective when set before the HIP runtime
# starts
_HW_QUEUES_WANTED = 16


def _hardware_queues():
    """GPU_MAX_HW_QUEUES is read once, when the HIP runtime starts.  Set it if nobody has; if the runtime is already up
    (the host application touched the GPU before importing this package) with fewer queues than the streams of a step
    need, say so: nothing fails, the step just serialises (measured: data-parallel step 50.7 -> 67.7 ms at 8 queues)."""
    import sys
    import warnings
    have = _os.environ.get("GPU_MAX_HW_QUEUES")
    torch_mod = sys.modules.get("torch")
    started = bool(torch_mod is not None and torch_mod.cuda.is_initialized())
    if have is None and not started:
        _os.environ["GPU_MAX_HW_QUEUES"] = str(_HW_QUEUES_WANTED)
        return
    try:
        n = int(have) if have is not None else 4      # the runtime's default
    except ValueError:
        n = 0
    if n < _HW_QUEUES_WANTED:
        when = "the HIP runtime was already running when lidog_amd was imported" if started else \
            f"GPU_MAX_HW_QUEUES={have} in the environment"
        warnings.warn(f"lidog_amd: {when}, with {n} hardware queues; the step keeps up to five streams busy (compute, "
                      f"weight gradients, coordinate maps, gradient buckets, RCCL) and streams that share a queue "
                      f"serialise. Export GPU_MAX_HW_QUEUES={_HW_QUEUES_WANTED} before the process first touches the GPU.",
                      RuntimeWarning, stacklevel=3)


_hardware_queues()


def _safe_mode():
    """LIDOG_DP_SAFE=1: the most conservative data-parallel schedule this package has -- what bench.py falls back to
    after an N > 1 run went silent (train_lidog.py:227-231's SyncBatchNorm + DDP, nothing else).  One communicator only
    (torch.distributed's own, for the statistics messages AND the gradient buckets), the buckets reduced on the compute
    stream after backward has ended (optim.GradientBuckets.deferred), no peer mailboxes, no downsample branches on
    side streams.  Set before the library reads its switches; explicit settings of the four switches are overridden."""
    if _os.environ.get("LIDOG_DP_SAFE") == "1":
        _os.environ.update(LIDOG_DP_TRANSPORT="torch", LIDOG_DP_BUCKETS="torch", LIDOG_PEER_ALLREDUCE="0",
                           LIDOG_SIDE_FORWARD="0", LIDOG_SIDE_BACKWARD="0")


_safe_mode()

from . import me, bev, losses, trunk  # noqa: E402,F401
from .minkunet import make_models  # noqa: E402

me.trunk_forward = trunk.trunk_forward   # the models hand their training-mode trunk pass to the executor

_models = make_models(me, bev.Encoder2D, bev.sparse2super)
MinkUNet34 = _models.MinkUNet34
MinkUNet34BEV = _models.MinkUNet34BEV
