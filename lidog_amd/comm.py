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
* on top of either, the statistics messages take the one-shot peer all-reduce of csrc/comm.hip (every rank pushes its
  vector into a mailbox in every peer's memory over its direct xGMI link and adds the N vectors in rank order) when
  every rank could open every other rank's mailbox and a start-up self-test gave the right sums
  (`LIDOG_PEER_ALLREDUCE=0` switches it off, `=1` also uses it in a one-rank group).
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
        self._init_peer()

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
        if self.peer is not None and t.numel() <= self.peer_max:
            _lib.call("lidog_peer_allreduce_f64", self.peer, _lib.ptr(t), t.numel())
        elif self.kind == "native":
            _lib.call("lidog_allreduce_f64", _lib.ptr(t), t.numel(), self.comm_bn)
        else:
            dist.all_reduce(t, group=self.group)

    def check(self):
        """raise if a wait of the peer all-reduce ever timed out (synchronises with the device: call it at epoch / run
        boundaries, not per step)"""
        if self.peer is not None and _lib.load().lidog_peer_status(self.peer) != 0:
            raise RuntimeError("lidog_amd.comm: a rank's statistics message did not arrive within the wait limit of the "
                               "peer all-reduce; results since then are invalid (LIDOG_PEER_ALLREDUCE=0 uses RCCL only)")

    # ---- one-shot peer all-reduce of the statistics messages (csrc/comm.hip)
    PEER_MAX_DOUBLES = 2 * (2 * 256 + 1)     # the joint conv1 + downsample message of a 256-channel block

    def _init_peer(self):
        """Mailboxes in every rank's memory, opened by every other rank through hipIpc handles that travel through the
        process group.  Used only if EVERY rank (a) could set it up and (b) got the right sums in a self-test with a
        short wait limit; anything else (no IPC between these processes, stores that do not become visible) leaves
        `peer` None and the statistics go through the communicator / torch.distributed as before."""
        self.peer, self.peer_max, self.peer_note = None, 0, "off"
        want = os.environ.get("LIDOG_PEER_ALLREDUCE", "auto")
        if want == "0" or not torch.cuda.is_available() or (self.world == 1 and want != "1"):
            return
        L = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        ok, comm, local = 1, ctypes.c_void_p(), ctypes.c_void_p()
        hb = L.lidog_peer_handle_bytes()
        handle = (ctypes.c_ubyte * hb)()
        nbytes = L.lidog_peer_mailbox_bytes(self.world, self.PEER_MAX_DOUBLES)
        if nbytes < 0 or L.lidog_peer_mailbox_alloc(nbytes, ctypes.byref(local), handle) != 0:
            ok, self.peer_note = 0, "mailbox allocation failed: " + L.lidog_last_error().decode()
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle) if ok else None, group=self.group)
        opened = []
        if ok and all(h is not None for h in handles):
            ptrs = (ctypes.c_void_p * self.world)()
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs[r] = local.value
                    continue
                p = ctypes.c_void_p()
                if L.lidog_peer_mailbox_open((ctypes.c_ubyte * hb)(*h), ctypes.byref(p)) != 0:
                    ok, self.peer_note = 0, f"cannot open rank {r}'s mailbox: " + L.lidog_last_error().decode()
                    break
                ptrs[r] = p.value
                opened.append(p.value)
            if ok and L.lidog_peer_comm_create(self.rank, self.world, self.PEER_MAX_DOUBLES, local, ptrs,
                                               ctypes.byref(comm)) != 0:
                ok, self.peer_note = 0, L.lidog_last_error().decode()
        else:
            ok = 0
        # self-test: every rank contributes f(rank, i); the sum is known.  Short wait limit (the ranks have just met in
        # the all_gather above), every rank must pass.
        if ok:
            L.lidog_peer_set_spin_limit(comm, 1 << 21)
            n = self.PEER_MAX_DOUBLES
            base = torch.arange(n, dtype=torch.float64, device=dev)
            want_sum = sum((r + 1) * 0.5 + base * (r + 3) for r in range(self.world))
            good = True
            for it in range(8):
                t = ((self.rank + 1) * 0.5 + base * (self.rank + 3)) * (it + 1)
                m = n if it % 2 == 0 else 193
                _lib.call("lidog_peer_allreduce_f64", comm, _lib.ptr(t), m)
                good = good and bool(torch.equal(t[:m], want_sum[:m] * (it + 1))) and bool(torch.equal(
                    t[m:], (((self.rank + 1) * 0.5 + base * (self.rank + 3)) * (it + 1))[m:]))
            if not good or L.lidog_peer_status(comm) != 0:
                ok, self.peer_note = 0, "self-test failed (wrong sums or a sender's flag never arrived)"
            L.lidog_peer_set_spin_limit(comm, 0)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 1:
            self.peer, self.peer_max, self.peer_note = comm.value, self.PEER_MAX_DOUBLES, "on"
        else:
            if self.peer_note == "off":
                self.peer_note = "another rank could not set it up"
            if comm.value:
                L.lidog_peer_comm_destroy(comm, 1)


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
        if tr.peer:
            L.lidog_peer_comm_destroy(tr.peer, 1)
    _TRANSPORTS.clear()
