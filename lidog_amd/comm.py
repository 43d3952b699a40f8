"""The thin RCCL wrapper that replaces Lightning DDP's collectives (train_lidog.py:227-231).

Two collectives exist on the path: the SyncBatchNorm statistics messages (<= 4 KB of doubles, 124 forward + 124
backward per step, on the dependent chain) and the gradient buckets (155 MB per step).  `transport(group)` decides
once per process group how they travel:

* ``native`` (default when the group's backend is nccl = RCCL): two communicators of this library's own
  (csrc/comm.hip, `lidog_comm_init_rank`; the unique ids travel through the torch process group once).  The statistics
  communicator's all-reduces are queued ON THE COMPUTE STREAM between the kernel that produces a message and the kernel
  that consumes it -- no second stream, no event pair per message --, the bucket communicator's on a stream of its
  own behind events of the compute and weight-gradient streams.  This is what lets the trunk executor
  (csrc/trunk.hip) run a data-parallel rank's whole pass from C.
* ``torch``: `torch.distributed` collectives of the group itself (any backend; gloo in the two-rank tests of this
  repository, which share one GPU where RCCL cannot put two ranks).  The executor reaches them through a host
  callback.

`LIDOG_DP_TRANSPORT=native|torch` overrides the choice.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib

_TRANSPORTS = {}


class Transport:
    def __init__(self, group):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        want = os.environ.get("LIDOG_DP_TRANSPORT", "")
        if want not in ("", "native", "torch"):
            raise ValueError(f"LIDOG_DP_TRANSPORT={want!r}: expected native or torch")
        backend = dist.get_backend(group)
        self.kind = want or ("native" if backend == "nccl" else "torch")
        # experiment switch: the gradient buckets through torch.distributed while the statistics stay native (or the
        # other way round with LIDOG_DP_TRANSPORT=torch LIDOG_DP_BUCKETS=native)
        self.bucket_kind = os.environ.get("LIDOG_DP_BUCKETS", "") or self.kind
        self.comm_bn = self.comm_grad = None
        self.stream = None
        if "native" in (self.kind, self.bucket_kind):
            if not torch.cuda.is_available():
                raise RuntimeError("the native RCCL transport needs a GPU")
            self.device = torch.device("cuda", torch.cuda.current_device())
            self.comm_bn = self._init_comm()
            self.comm_grad = self._init_comm()
            self.stream = torch.cuda.Stream(device=self.device)
            self.raw_stream = self.stream.cuda_stream

    def _init_comm(self):
        """one RCCL communicator over the ranks of the group; rank 0's unique id reaches the others through the group"""
        L = _lib.load()
        nbytes = L.lidog_comm_unique_id_bytes()
        uid = (ctypes.c_ubyte * nbytes)()
        if self.rank == 0:
            if L.lidog_comm_unique_id(uid) != 0:
                raise RuntimeError(L.lidog_last_error().decode())
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor(list(uid), dtype=torch.uint8, device=self.device if on_gpu else "cpu")
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src, group=self.group)
        uid = (ctypes.c_ubyte * nbytes)(*t.cpu().tolist())
        comm = ctypes.c_void_p()
        if L.lidog_comm_init_rank(uid, self.world, self.rank, ctypes.byref(comm)) != 0:
            raise RuntimeError(L.lidog_last_error().decode())
        return comm.value

    def allreduce_f64(self, t):
        """sum over the ranks, in order on the current stream (SyncBatchNorm statistics of the operator path)"""
        if self.kind == "native":
            _lib.call("lidog_allreduce_f64", _lib.ptr(t), t.numel(), self.comm_bn)
        else:
            dist.all_reduce(t, group=self.group)


def transport(group=None):
    """the Transport of a process group (None = the default group), created on first use by every rank together"""
    if group is not None and group is dist.group.WORLD:
        group = None
    key = id(group) if group is not None else None
    tr = _TRANSPORTS.get(key)
    if tr is None:
        tr = _TRANSPORTS[key] = Transport(group)
    return tr


def reset():
    """forget every transport (after dist.destroy_process_group(); the communicators are released)"""
    L = _lib.load()
    for tr in _TRANSPORTS.values():
        for c in (tr.comm_bn, tr.comm_grad):
            if c:
                L.lidog_comm_destroy(c)
    _TRANSPORTS.clear()
