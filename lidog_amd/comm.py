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
* on top of either, with `LIDOG_PEER_ALLREDUCE=1` the statistics messages take the one-shot peer all-reduce of
  csrc/comm.hip (every rank pushes its vector into a mailbox in every peer's memory over its direct xGMI link and adds
  the N vectors in rank order) when every rank could open every other rank's mailbox and a start-up self-test gave the
  right sums.  `=probe` sets it up, self-tests it and lets bench.py MEASURE it next to RCCL; by default it is not even
  set up (it has never run between two GPUs).
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
            self.peer_bind()
            _lib.call("lidog_peer_allreduce_f64", self.peer, _lib.ptr(t), t.numel())
        elif self.kind == "native":
            _lib.call("lidog_allreduce_f64", _lib.ptr(t), t.numel(), self.comm_bn)
        else:
            dist.all_reduce(t, group=self.group)

    def peer_bind(self, comm=None):
        """the peer communicator's calls must all be queued on ONE stream (its two mailbox slots are reused in stream
        order): follow the caller when torch's current stream has changed (the library waits for the old stream then;
        a no-op otherwise)"""
        comm = comm or self.peer
        if comm is not None and _lib.load().lidog_peer_rebind_stream(comm, _lib.stream()) != 0:
            raise RuntimeError(_lib.load().lidog_last_error().decode())

    def check(self):
        """Raise -- on EVERY rank together -- if a wait of the peer all-reduce ever timed out or a rank fell out of
        step (collective: every rank of the group must call it; synchronises with the device, so call it at epoch /
        run boundaries or every few dozen steps, not per step).  After a failure no later call waits for anything
        (csrc/comm.hip), so the ranks do reach this point."""
        comm = self.peer or self.peer_probe
        if comm is None:
            return
        st = _lib.load().lidog_peer_status(comm)
        st = 3 if st < 0 else st
        if self.world > 1:
            on_gpu = dist.get_backend(self.group) == "nccl"
            t = torch.tensor([st], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            st = int(t.item())
        if st != 0:
            what = {1: "a rank's statistics message did not arrive within the wait limit",
                    2: "a rank was ahead of another one (the ranks stopped making the same calls)"}.get(st, "status unreadable")
            raise RuntimeError(f"lidog_amd.comm: peer all-reduce failed on some rank: {what}; results since then are "
                               "invalid (LIDOG_PEER_ALLREDUCE=0 uses RCCL only)")

    # ---- one-shot peer all-reduce of the statistics messages (csrc/comm.hip)
    PEER_MAX_DOUBLES = 2 * (2 * 256 + 1)     # the joint conv1 + downsample message of a 256-channel block

    def _init_peer(self):
        """Mailboxes in every rank's memory, opened by every other rank through hipIpc handles that travel through the
        process group.  Kept only if EVERY rank (a) could set it up and (b) got the right sums in a self-test with a
        short wait limit; anything else (no IPC between these processes, stores that do not become visible) leaves
        nothing behind and the statistics go through the communicator / torch.distributed.

        LIDOG_PEER_ALLREDUCE: "1" = the step's statistics messages take it (where the set-up succeeded on every rank);
        "probe" = it is set up and self-tested but only MEASURED (`peer_probe`: bench.py times it next to RCCL);
        unset / "0" = not even set up: the path has never run between two GPUs (one-GPU boxes only), its self-test
        already stores through hipIpc mappings of another device's memory, and the first multi-GPU run of this code must
        not depend on that.  In a one-rank group only "1" sets it up.
        Fault injection (tests): LIDOG_PEER_FAULT="open:<rank>" makes that rank fail to open its peers' mailboxes;
        "skipflag:<rank>:<k>" makes that rank's k-th call after the self-test raise no flags."""
        self.peer, self.peer_probe, self.peer_max, self.peer_note = None, None, 0, "off"
        want = os.environ.get("LIDOG_PEER_ALLREDUCE", "0")
        if want not in ("1", "probe") or not torch.cuda.is_available() or (self.world == 1 and want != "1"):
            return
        fault = os.environ.get("LIDOG_PEER_FAULT", "").split(":")
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
                if fault[0] == "open" and int(fault[1]) == self.rank:
                    ok, self.peer_note = 0, f"cannot open rank {r}'s mailbox: injected fault (LIDOG_PEER_FAULT)"
                    break
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
        # One round of agreement BEFORE the self-test: a rank that could not set up must not leave the others running
        # eight all-reduces against a partner that never writes (eight wait limits of seconds each).
        ok = self._all_min(ok, dev)
        # self-test: every rank contributes f(rank, i); the sum is known.  Short wait limit (the ranks have just met
        # above), every rank must pass.
        if ok:
            L.lidog_peer_set_spin_limit(comm, 1 << 21)
            n = self.PEER_MAX_DOUBLES
            base = torch.arange(n, dtype=torch.float64, device=dev)
            want_sum = sum((r + 1) * 0.5 + base * (r + 3) for r in range(self.world))
            good = True
            self.peer_bind(comm.value)
            for it in range(8):
                t = ((self.rank + 1) * 0.5 + base * (self.rank + 3)) * (it + 1)
                m = n if it % 2 == 0 else 193
                _lib.call("lidog_peer_allreduce_f64", comm, _lib.ptr(t), m)
                good = good and bool(torch.equal(t[:m], want_sum[:m] * (it + 1))) and bool(torch.equal(
                    t[m:], (((self.rank + 1) * 0.5 + base * (self.rank + 3)) * (it + 1))[m:]))
            if not good or L.lidog_peer_status(comm) != 0:
                ok, self.peer_note = 0, "self-test failed (wrong sums or a sender's flag never arrived)"
            L.lidog_peer_set_spin_limit(comm, int(os.environ.get("LIDOG_PEER_SPIN_LIMIT", "0")))
            ok = self._all_min(ok, dev)
        if ok:
            if fault[0] == "skipflag" and int(fault[1]) == self.rank:
                L.lidog_peer_inject_skip_flag(comm, L.lidog_peer_calls(comm) + int(fault[2]))
            self.peer_max = self.PEER_MAX_DOUBLES
            if want == "1":
                self.peer, self.peer_note = comm.value, "on"
            else:
                self.peer_probe = comm.value
                self.peer_note = "set up and self-tested, measured only (LIDOG_PEER_ALLREDUCE=1 switches it on)"
        else:
            if self.peer_note == "off":
                self.peer_note = "another rank could not set it up"
            # nothing of a half-built set-up stays behind: the communicator owns the mailbox and the mappings once it
            # exists; before that they are released one by one
            if comm.value:
                L.lidog_peer_comm_destroy(comm, 1)
            else:
                for p in opened:
                    L.lidog_peer_mailbox_close(p)
                if local.value:
                    L.lidog_peer_mailbox_free(local)

    def _all_min(self, ok, dev):
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item())


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
    """release every transport: this library's RCCL communicators and the peer mailboxes.  Call it on every rank once
    all collectives are done (barrier + device synchronisation), before dist.destroy_process_group()."""
    L = _lib.load()
    for tr in _TRANSPORTS.values():
        for c in (tr.comm_bn, tr.comm_grad):
            if c:
                L.lidog_comm_destroy(c)
        if tr.peer or tr.peer_probe:
            L.lidog_peer_comm_destroy(tr.peer or tr.peer_probe, 1)
    _TRANSPORTS.clear()
