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
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import me, bev, losses, trunk  # noqa: E402,F401
from .minkunet import make_models  # noqa: E402

me.trunk_forward = trunk.trunk_forward   # the models hand their training-mode trunk pass to the executor

_models = make_models(me, bev.Encoder2D, bev.sparse2super)
MinkUNet34 = _models.MinkUNet34
MinkUNet34BEV = _models.MinkUNet34BEV
