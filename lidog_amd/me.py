"""MinkowskiEngine-compatible operator API on MI355X (HIP kernels behind include/lidog_amd.h).

Mirrors the surface of MinkowskiEngine 0.5.4 that LiDOG's models and pipelines use
(SURVEY.md 8(b)); reference call sites:
  SparseTensor(coordinates=..., features=...)        utils/pipelines/trainer_lighting_2d.py:151
  MinkowskiConvolution / ...Transpose                utils/models/minkunet_bev.py:57-123
  MinkowskiBatchNorm(.bn) / MinkowskiSyncBatchNorm   minkunet_bev.py:60,406-408; train_lidog.py:228
  MinkowskiReLU(inplace=True), cat, +=               minkunet_bev.py:124,337; resnet_block.py:52
  utils.kaiming_normal_                              minkunet_bev.py:404
  modules.resnet_block.BasicBlock                    minkunet_bev.py:4,425-439

`install_as_minkowski_engine()` registers this module as `MinkowskiEngine` so the reference's
`import MinkowskiEngine as ME` resolves to it.  Everything runs on the GPU through the C ABI; there
is no CPU path (tensors on the CPU raise).
"""
import ctypes
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, call_on, ptr

TILE_ROWS = 128  # GM_TM of csrc/sconv.hip


def kernel_offsets(kernel_size, tensor_stride, dilation=1):
    """[K,3] int32 (x,y,z), index x fastest; odd sizes centred, even sizes start at 0."""
    k = int(kernel_size)
    if k % 2 == 1:
        r = [(-(k // 2) + i) * tensor_stride * dilation for i in range(k)]
    else:
        r = [i * tensor_stride * dilation for i in range(k)]
    return np.array([(x, y, z) for z in r for y in r for x in r], dtype=np.int32)


def _tiles_host(k_off_host, skip_k=-1):
    """tile descriptors (tile_k, tile_row0, tile_rows) for 128-row tiles that never straddle an offset:
    (int32 [3, n_tiles] numpy array, n_tiles).  `skip_k`: an offset left out.
    Launch order: tiles at the same relative position of their offset segment run together.  Pairs are sorted by output
    row inside a segment, so these tiles gather (nearly) the same feature rows for different offsets while they are
    still in L2 instead of re-fetching them K times from HBM.  (Host code of the library, csrc/hostprep.hip: the numpy
    formulation cost ~3 ms of python per training step together with the weight-gradient items below.)"""
    k_off = np.ascontiguousarray(k_off_host, dtype=np.int64)
    K = k_off.shape[0] - 1
    cap = int(k_off[-1] - k_off[0]) // TILE_ROWS + K + 1
    out = np.empty(3 * cap, dtype=np.int32)
    n = _lib.load().lidog_tiles_host(k_off.ctypes.data, K, int(skip_k), TILE_ROWS, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("lidog_tiles_host: capacity")
    return out[:3 * n].reshape(3, n), int(n)


def _tiles(k_off_host, device, skip_k=-1):
    desc, total = _tiles_host(k_off_host, skip_k)
    return torch.from_numpy(desc).to(device), total


class _Arena:
    """Host-built int32 tables (tile descriptors, weight-gradient work items) of many maps, shipped to the device
    in ONE copy; `views()` hands the pieces back in order."""

    def __init__(self):
        self.parts = []

    def add(self, arr):
        self.parts.append(np.ascontiguousarray(arr, dtype=np.int32))
        return len(self.parts) - 1

    def ship(self, device):
        flat = np.concatenate([a.ravel() for a in self.parts]) if self.parts else np.zeros(0, np.int32)
        dev = torch.from_numpy(flat).to(device)
        out, off = [], 0
        for a in self.parts:
            out.append(dev[off:off + a.size].view(a.shape))
            off += a.size
        return dev, out


class _CoordMap:
    __slots__ = ("coords", "keys", "vals", "cap", "n", "bits", "box")

    def __init__(self, coords, keys, vals, cap):
        self.coords, self.keys, self.vals, self.cap, self.n = coords, keys, vals, cap, coords.shape[0]
        self.bits = self.box = None    # occupancy bitmap over the bounding box (CoordinateManager._bitmap)


class KernelMap:
    """Rule book of one (in map, out map, kernel) triple; see include/lidog_amd.h:lidog_kernel_map_pairs."""

    def __init__(self, K, n_in, n_out, k_off, k_off_host, pair_in, pair_out, pos_out, pos_in, nbr, tiles=None):
        self.K, self.n_in, self.n_out = K, n_in, n_out
        self.k_off, self.k_off_host = k_off, k_off_host
        self.P = int(k_off_host[-1])
        self.pair_in, self.pair_out, self._pos_out, self._pos_in, self.nbr = pair_in, pair_out, pos_out, pos_in, nbr
        self.tiles, self.n_tiles = tiles if tiles is not None else _tiles(k_off_host, pair_in.device)
        self._rows = {}

    def _pos_table(self, side):
        """[K, n] pair position of (offset, row) or -1.  Written with the rule book for K <= 27; a 5^3 map (the Cin = 1
        stem never walks it by row) gets it on first request, from the pair lists (not on the training path)."""
        name = "_pos_" + side
        t = getattr(self, name)
        if t is None:
            n = self.n_out if side == "out" else self.n_in
            rows = (self.pair_out if side == "out" else self.pair_in).long()
            dev = rows.device
            t = torch.full((self.K, n), -1, dtype=torch.int32, device=dev)
            counts = torch.tensor(np.diff(np.asarray(self.k_off_host, dtype=np.int64)), device=dev)
            ks = torch.repeat_interleave(torch.arange(self.K, device=dev), counts)
            t[ks, rows] = torch.arange(self.P, dtype=torch.int32, device=dev)
            setattr(self, name, t)
        return t

    pos_out = property(lambda self: self._pos_table("out"))
    pos_in = property(lambda self: self._pos_table("in"))

    def sorted(self):
        """(perm, wave_masks, tile_order) of lidog_kernel_map_sorted -- the rows sorted by neighbour mask, for the
        output-stationary convolution (csrc/sconv_os.hip) -- or None where that kernel does not apply: only 3^3 maps
        of a coordinate map onto itself (symmetric: one sorted order serves forward and data gradient), only sparse
        ones (LIDOG_SCONV_OS_DENSITY pairs per row at most, default 6: denser maps fill the two-pass path's tiles
        better than they fill these; measured at tensor strides 1 / 2 vs 4 and up of the bench scans) and only large
        ones (LIDOG_SCONV_OS_MIN_TILES 128-row tiles, default 1500: a tile walks 5 to 27 offsets one after the other,
        so a grid of a few hundred tiles ends on its longest ones -- 8 k-point scans ran 2.4 x slower with it).
        LIDOG_SCONV_OS=0 switches it off, =2 takes it for every symmetric 3^3 map."""
        if "_sorted" not in self.__dict__:
            self._sorted = None
            ok = _SCONV_OS and self.K == 27 and self.n_in == self.n_out and self.nbr is not None and self.n_out > 0
            if ok and (_SCONV_OS == 2 or (self.P <= _SCONV_OS_DENSITY * self.n_out and
                                          self.n_out >= 128 * _SCONV_OS_MIN_TILES)):
                n, dev = self.n_out, self.nbr.device
                pad = (n + 127) // 128 * 128
                perm = torch.empty(pad, dtype=torch.int32, device=dev)
                wmask = torch.empty(pad // 32, dtype=torch.int32, device=dev)
                order = torch.empty(pad // 128, dtype=torch.int32, device=dev)
                ws = torch.empty(_lib.load().lidog_kernel_map_sorted_ws(n), dtype=torch.uint8, device=dev)
                call("lidog_kernel_map_sorted", ptr(self.nbr), n, self.K, ptr(self.k_off), ptr(perm), ptr(wmask),
                     ptr(order), ptr(ws), ws.numel())
                self._sorted = (perm, wmask, order)
        return self._sorted

    def rows(self, side):
        """(row_ptr int32 [n+1], row_list int32 [P]) of the output ("out") or input ("in") rows: the pair positions
        of every row in ascending offset order -- what the reduction pass walks (include/lidog_amd.h:
        lidog_kernel_map_rows).  Built with the map when it is prepared ahead of time, else on first use."""
        key = side
        if key not in self._rows:
            pos, n = (self.pos_out, self.n_out) if side == "out" else (self.pos_in, self.n_in)
            dev = pos.device
            row_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            row_list = torch.empty(max(self.P, 1), dtype=torch.int32, device=dev)
            ws = torch.empty((n + 1 + 1023) // 1024 + 1, dtype=torch.int32, device=dev)
            call("lidog_kernel_map_rows", ptr(pos), n, self.K, -1, ptr(row_ptr),
                 ptr(row_list), ptr(ws))
            self._rows[key] = (row_ptr, row_list)
        return self._rows[key]


class _IdentityMap:
    """tile descriptors / pair lists of a 1x1 convolution over n rows (K = 1, in row == out row)"""

    def __init__(self, n, device, tiles=None):
        self.K, self.n_in, self.n_out, self.P = 1, n, n, n
        self.k_off_host = [0, n]
        self.k_off = torch.tensor([0, n], dtype=torch.int64, device=device)
        self.rows = torch.arange(n, dtype=torch.int32, device=device)
        self.tiles, self.n_tiles = tiles if tiles is not None else _tiles(self.k_off_host, device)


_SIDE_STREAMS = {}
# output-stationary 3^3 convolution (KernelMap.sorted): 0 = off, 1 = sparse symmetric maps (default), 2 = every one
_SCONV_OS = int(os.environ.get("LIDOG_SCONV_OS", "1"))
_SCONV_OS_DENSITY = float(os.environ.get("LIDOG_SCONV_OS_DENSITY", "6.0"))
_SCONV_OS_MIN_TILES = int(os.environ.get("LIDOG_SCONV_OS_MIN_TILES", "1500"))
_OS_HINT = {}     # kernel-map key -> the map of the previous batch took the output-stationary kernel


def _side_stream(device):
    """the stream coordinate maps are prepared on when they are built ahead of time (CoordinateManager.prepare)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class _WgradLane:
    """Second stream of the backward pass.  A weight gradient depends only on tensors that exist when its
    convolution's backward starts and nothing in the backward chain depends on it, so it is queued on a side
    stream.  It is forked BEHIND the data gradient's GEMM of its layer: two matrix kernels next to each other gain
    nothing (same pipe), but the weight gradient then runs next to what follows that GEMM on the main stream -- the
    per-row reduction, the BatchNorm backward kernels of the next layer, their microsecond-sized follow-ups and, in
    data-parallel runs, the SyncBatchNorm collectives -- all bandwidth- or latency-bound.  Measured on one GPU
    (bs 4): 77.7 -> 81.2 scans/s; forked BEFORE the GEMM (LIDOG_BACKWARD_OVERLAP=1) 79.9, with every data-gradient
    GEMM sharing the chip with a weight gradient (its timed duration grows by 45 %, against 6 % behind the GEMM,
    where only the tail of the previous layer's weight gradient can reach into it).
    Only used when the gradient is written straight into the optimiser's flat buffer (me._grad_out): a fresh tensor
    would be touched by autograd on the main stream right after backward() returns it.
    Tensors read on the side stream are kept alive until the join (also keeps autograd from accumulating into
    them in place); the join is an engine callback at the end of the backward pass."""

    # LIDOG_BACKWARD_OVERLAP: unset / 2 = behind the GEMM (default), 1 = before it, 0 = everything on one stream
    _env = os.environ.get("LIDOG_BACKWARD_OVERLAP", "2")
    enabled = _env != "0"
    mode = 1 if _env == "1" else 2
    _lanes = {}

    @classmethod
    def active(cls):
        return cls.enabled

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.raw = self.stream.cuda_stream    # hipStream_t: kernels are launched on it without switching torch's stream
        self.keep = []
        self.pending = False

    @classmethod
    def get(cls, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in cls._lanes:
            cls._lanes[key] = cls(torch.device("cuda", key))
        return cls._lanes[key]

    def fork(self, *tensors):
        """side stream waits for everything queued on the current stream so far; returns the side stream"""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.keep.extend(tensors)
        if not self.pending:
            self.pending = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.join)
            except RuntimeError:   # not inside a backward pass: the caller joins
                pass
        return self.stream

    def join(self):
        if self.pending:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self.pending = False
            self.keep = []


def set_backward_overlap(on, mode=2):
    """weight gradients on a second stream (True; mode 2 = forked behind the data gradient's GEMM, 1 = before it)
    or in line on the current stream (False)"""
    _WgradLane.enabled = bool(on)
    _WgradLane.mode = 1 if mode == 1 else 2


def wgrad_lane(device):
    """the lane if weight gradients may still be in flight on it, else None (lidog_amd.trainer orders its
    gradient all-reduces after it)"""
    key = device.index if device.index is not None else torch.cuda.current_device()
    lane = _WgradLane._lanes.get(key)
    return lane if lane is not None and lane.pending else None


class CoordinateManager:
    """Coordinate maps keyed by tensor stride, kernel maps cached per (in, out, kernel, dilation)."""

    def __init__(self, device):
        self.device = device
        self.maps = {}
        self.kmaps = {}
        self.identity = {}
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self.batch_size = None
        # (map key, Cin, Cout) of every convolution that used this manager, in call order; a later batch of the
        # same network can be prepared ahead of time from it (CoordinateManager.prepare)
        self.trace = []
        self.uniq = None
        self._ready = None      # event on the side stream once prepare() has finished
        self._owned = [self.err]  # device tensors created here (handed to the consumer stream by handover())
        self._nbr_tables = {}     # kernel-map key -> neighbour table, from the moment its kernel has been queued

    def _own(self, *tensors):
        self._owned.extend(tensors)
        return tensors[0] if len(tensors) == 1 else tensors

    @classmethod
    def prepare(cls, coordinates, trace, ready_event=None):
        """Build, on a side stream, the coordinate manager of `coordinates` with every coordinate map, kernel map,
        tile list and weight-gradient work list named in `trace` (the .trace of a manager that went through the
        same network).  The host synchronises with the side stream only, so the call overlaps with whatever is
        queued on the current stream (the previous training step).  `ready_event`: event after which
        `coordinates` is valid; None = everything queued on the current stream so far.
        Pass the result as ME.SparseTensor(features, coordinates=..., coordinate_manager=cm)."""
        _lib.require_gpu(coordinates, "coordinates")
        dev = coordinates.device
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        if ready_event is not None:
            side.wait_event(ready_event)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            cm = cls(dev)
            cm.uniq, _ = cm.insert(coordinates)
            cm._own(coordinates)
            cm.prefetch(trace)
            cm._ready = torch.cuda.Event()
            cm._ready.record(side)
        return cm

    def handover(self):
        """make the current stream wait for prepare() and tell the allocator that it reads the map tensors"""
        if self._ready is None:
            return
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._ready)
        for t in self._owned:
            t.record_stream(cur)
        self._ready = None

    def prefetch(self, trace):
        """Build every map of `trace` that is missing with ONE host synchronisation for all kernel maps and ONE
        host-to-device copy for all tile / work-item tables (the lazy path costs one of each per map, each with
        the GPU idle while the host prepares the tables)."""
        want, shapes = [], {}
        for key, cin, cout in trace:
            if key not in shapes:
                shapes[key] = []
                want.append(key)
            if (cin, cout) not in shapes[key]:
                shapes[key].append((cin, cout))
        pending = []
        for key in want:
            if key[0] == "identity":
                continue
            s_in, s_out, ksize, dil = key
            if key in self.kmaps:
                continue
            if s_out != s_in:
                self.stride(s_in, s_out)
            pending.append((key, self._kernel_map_launch(key)))
        # the error word rides along: k_bitmap_set flags a coordinate outside the box its bitmap was sized for
        hosts = torch.cat([pd[3] for _, pd in pending] + [self.err.long()]).tolist() if pending else []
        if pending and hosts[-1] != 0:
            self._check(int(hosts[-1]))
        arena, todo, off = _Arena(), [], 0
        for key, pd in pending:
            K = pd[0]
            k_off_host = hosts[off:off + K + 1]
            off += K + 1
            desc, n_tiles = _tiles_host(k_off_host)
            todo.append(("kmap", key, pd, k_off_host, arena.add(desc), n_tiles))
        for key in want:
            if key[0] == "identity":
                n = self.maps[key[1]].n
                if n not in self.identity:
                    desc, n_tiles = _tiles_host([0, n])
                    todo.append(("identity", n, None, [0, n], arena.add(desc), n_tiles))
        # weight-gradient work items of every (map, Cin, Cout) seen
        items_todo = []
        k_off_of = {t[1]: t[3] for t in todo}
        for key in want:
            if key[0] == "identity":
                ident, koh = self.maps[key[1]].n, None
                koh = k_off_of.get(ident, [0, ident])
            elif key in k_off_of:
                koh = k_off_of[key]
            else:
                koh = self.kmaps[key].k_off_host
            seen = set()
            for cin, cout in shapes[key]:
                chunk = _wgrad_chunk(koh, cin, cout)
                if chunk in seen:
                    continue
                seen.add(chunk)
                items, total, item_off = _wgrad_items_host(koh, chunk)
                items_todo.append((key, chunk, arena.add(items), total, arena.add(item_off)))
        dev, views = arena.ship(self.device)
        self._own(dev)
        for kind, key, pd, k_off_host, slot, n_tiles in todo:
            if kind == "kmap":
                self.kmaps[key] = self._kernel_map_finish(pd, k_off_host, (views[slot], n_tiles), key)
            else:
                self.identity[key] = _IdentityMap(key, self.device, (views[slot], n_tiles))
                self._own(self.identity[key].k_off, self.identity[key].rows)
        for key, chunk, islot, total, oslot in items_todo:
            m = self.identity[self.maps[key[1]].n] if key[0] == "identity" else self.kmaps[key]
            m.__dict__.setdefault("_wgrad_items", {})[chunk] = (views[islot], total, views[oslot])

    def _check(self, code=None):
        code = int(self.err.item()) if code is None else code
        if code == 2:
            raise RuntimeError("occupancy bitmap: a coordinate lies outside the bounding box the bitmap was sized for "
                               "(kernel maps built through it would drop neighbours); LIDOG_MAP_BITMAPS=0 disables them")
        if code != 0:
            raise ValueError("coordinates out of the supported range: |x|,|y|,|z| <= 65535, 0 <= batch <= 4095")

    def insert(self, coords):
        _lib.require_gpu(coords, "coordinates")
        coords = coords.contiguous()
        n = coords.shape[0]
        cap = _lib.load().lidog_hash_capacity(n)
        keys = self._own(torch.empty(cap, dtype=torch.int64, device=self.device))
        vals = self._own(torch.empty(cap, dtype=torch.int32, device=self.device))
        first = torch.empty(n, dtype=torch.int32, device=self.device)
        # unique rows, error flag, largest batch index and the bounding box of the voxels (for the occupancy bitmaps) come
        # out of the insert kernel itself: ONE read-back, no reduction kernels of torch's on this stream
        info = torch.empty(9, dtype=torch.int64, device=self.device)
        call("lidog_coords_insert_info", ptr(coords), n, ptr(keys), ptr(vals), cap, ptr(first), ptr(info),
             ptr(self.err))
        stats = info.tolist()
        self.bounds = (tuple(stats[3:6]), tuple(stats[6:9])) if (n >= _BITMAP_MIN_ROWS and _BITMAPS) else None
        if stats[1] != 0:
            self._check(1)
        self.batch_size = int(stats[2]) + 1 if n else 0
        uniq = inv = None
        if stats[0] != n:  # duplicates: keep the first occurrence, rows in first-occurrence order
            m = int(stats[0])
            uniq_full = torch.empty(n, dtype=torch.int32, device=self.device)
            inv = torch.empty(n, dtype=torch.int32, device=self.device)
            ws = torch.empty(n + 2048, dtype=torch.int32, device=self.device)
            call("lidog_coords_compact", ptr(first), n, ptr(keys), ptr(vals), cap, ptr(coords), ptr(uniq_full),
                 ptr(inv), ptr(ws))
            uniq = uniq_full[:m]
            coords = coords[uniq.long()].contiguous()
            self._own(uniq_full, inv, coords)
        self.maps[1] = _CoordMap(coords, keys, vals, cap)
        return uniq, inv

    def stride(self, s_in, s_out):
        if s_out in self.maps:
            return self.maps[s_out]
        src = self.maps[s_in]
        n = src.n
        cap = _lib.load().lidog_hash_capacity(n)
        keys = self._own(torch.empty(cap, dtype=torch.int64, device=self.device))
        vals = self._own(torch.empty(cap, dtype=torch.int32, device=self.device))
        p2c = torch.empty(n, dtype=torch.int32, device=self.device)
        out = self._own(torch.empty((n, 4), dtype=torch.int32, device=self.device))
        n_out = torch.zeros(1, dtype=torch.int64, device=self.device)
        ws = torch.empty(2 * n + 2048, dtype=torch.int32, device=self.device)
        call("lidog_coords_stride", ptr(src.coords), n, int(s_out), ptr(keys), ptr(vals), cap, ptr(p2c), ptr(out),
             ptr(n_out), ptr(ws), ptr(self.err))
        m = int(n_out.item())
        self.maps[s_out] = _CoordMap(out[:m], keys, vals, cap)  # out[:m] stays contiguous (leading rows)
        return self.maps[s_out]

    def _bitmap(self, s):
        """occupancy bitmap of the coordinate map of tensor stride s over the batch's bounding box (bits, box) or
        (None, None): no bounds known, or the box would take more than _BITMAP_MAX_BYTES"""
        cmap = self.maps[s]
        if cmap.box is not None:
            return cmap.bits, cmap.box
        cmap.box = ()
        # small maps: the probes are cheap and the bitmap's zero-fill + set kernels are not (8 k-point scans: +0.3 ms)
        if not _BITMAPS or getattr(self, "bounds", None) is None or cmap.n < _BITMAP_MIN_ROWS:
            return None, ()
        lo, hi = self.bounds
        x0 = [(int(v) // s) * s for v in lo]            # python floor division: toward -inf, as the strided maps do
        nn = [(int(h) // s) * s // s - a // s + 1 for h, a in zip(hi, x0)]
        words = _lib.load().lidog_bitmap_words(nn[0], nn[1], nn[2], self.batch_size, _BITMAP_MAX_BYTES)
        if words < 0:
            return None, ()
        bits = self._own(torch.zeros(words, dtype=torch.int32, device=self.device))
        box = (x0[0], x0[1], x0[2], nn[0], nn[1], nn[2], s, self.batch_size)
        call("lidog_bitmap_set", ptr(cmap.coords), cmap.n, *box, ptr(bits), ptr(self.err))
        cmap.bits, cmap.box = bits, box
        return bits, box

    def kernel_map(self, s_in, s_out, kernel_size, dilation=1):
        key = (s_in, s_out, kernel_size, dilation)
        if key in self.kmaps:
            return self.kmaps[key]
        if s_out != s_in:
            self.stride(s_in, s_out)
        pd = self._kernel_map_launch(key)
        k_off_host = torch.cat([pd[3], self.err.long()]).tolist()  # one synchronisation per kernel map on this (lazy) path
        if k_off_host.pop() != 0:
            self._check()
        self.kmaps[key] = self._kernel_map_finish(pd, k_off_host, None, key)
        return self.kmaps[key]

    def _kernel_map_launch(self, key):
        """queue the neighbour search and the rule-book compaction of one kernel map (no host synchronisation)"""
        s_in, s_out, kernel_size, dilation = key
        cin = self.maps[s_in]
        cout = cin if s_out == s_in else self.maps[s_out]
        offs = kernel_offsets(kernel_size, s_in, dilation)
        K = offs.shape[0]
        n_in, n_out = cin.n, cout.n
        nbr = torch.empty((K, n_out), dtype=torch.int32, device=self.device)
        # a 3^3 map of a coordinate map onto itself whose 5^3 map exists already (the stem's, built first: the trace is in
        # forward order): its 27 offsets are rows of that table -- no probes at all (lidog_kernel_map_subset)
        big = self._nbr_tables.get((s_in, s_out, 5, dilation)) if (_SUBSET_MAPS and kernel_size == 3 and s_in == s_out) else None
        bits, box = (None, None) if big is not None else self._bitmap(s_in)
        if big is not None:
            call("lidog_kernel_map_subset", ptr(big), n_out, 125, _SUBSET_3_OF_5.ctypes.data, K, ptr(nbr))
        elif bits is not None:
            call("lidog_kernel_map_bits", ptr(cout.coords), n_out, ptr(cin.keys), ptr(cin.vals), cin.cap,
                 offs.ctypes.data, K, ptr(bits), *box, ptr(nbr))
        else:
            call("lidog_kernel_map", ptr(cout.coords), n_out, ptr(cin.keys), ptr(cin.vals), cin.cap,
                 offs.ctypes.data, K, ptr(nbr))
        k_off = torch.empty(K + 1, dtype=torch.int64, device=self.device)
        pair_in = torch.empty(n_out * K, dtype=torch.int32, device=self.device)
        pair_out = torch.empty(n_out * K, dtype=torch.int32, device=self.device)
        # position tables: what the per-row lists are built from (3^3 and 2^3 maps) and what the dense-table reduction
        # of odd channel counts walks; a 5^3 map (the stem: straight from the neighbour table) never needs them
        # ... nor does a map whose rows were sorted for the output-stationary kernel the last time a batch went through
        # this network (_OS_HINT: the decision needs the pair count, which this batch does not have yet; if it comes out
        # differently the tables are rebuilt from the pair lists on first use, KernelMap._pos_table)
        by_row = K <= 27 and not (K == 27 and s_in == s_out and _OS_HINT.get(key, False))
        pos_out = torch.empty((K, n_out), dtype=torch.int32, device=self.device) if by_row else None
        pos_in = torch.empty((K, n_in), dtype=torch.int32, device=self.device) if by_row else None
        nbp = (n_out + 1023) // 1024
        ws = torch.empty((nbp + 1) * K + K + 2, dtype=torch.int32, device=self.device)
        call("lidog_kernel_map_pairs", ptr(nbr), n_out, n_in, K, ptr(k_off), ptr(pair_in), ptr(pair_out),
             ptr(pos_out), ptr(pos_in), ptr(ws))
        self._own(*[t for t in (nbr, k_off, pair_in, pair_out, pos_out, pos_in) if t is not None])
        self._nbr_tables[key] = nbr
        return (K, n_in, n_out, k_off, pair_in, pair_out, pos_out, pos_in, nbr)

    def _kernel_map_finish(self, pd, k_off_host, tiles, key=None):
        K, n_in, n_out, k_off, pair_in, pair_out, pos_out, pos_in, nbr = pd
        P = int(k_off_host[-1])
        m = KernelMap(K, n_in, n_out, k_off, k_off_host, pair_in[:P], pair_out[:P], pos_out, pos_in, nbr, tiles)
        # per-row lists for the reduction passes that will use this map (3^3: forward and data gradient; 2^3 stride 2:
        # the strided convolution's forward and the transposed convolution's data gradient, both over the coarse rows)
        if K == 27:
            srt = m.sorted()
            if key is not None:
                _OS_HINT[key] = srt is not None
            if srt is not None:
                self._own(*srt)      # nobody walks this map by row: no per-row lists (built on first use if asked for)
            else:
                self._own(*m.rows("out"), *m.rows("in"))
        elif K == 8:
            self._own(*m.rows("out"))
        return m

    def identity_map(self, n):
        if n not in self.identity:
            self.identity[n] = _IdentityMap(n, self.device)
        return self.identity[n]


class SparseTensor:
    def __init__(self, features=None, coordinates=None, coordinate_manager=None, coordinate_map_key=None,
                 tensor_stride=1, **unsupported):
        if unsupported:   # e.g. quantization_mode, minkowski_algorithm: not part of the LiDOG hot path
            raise TypeError(f"lidog_amd.me.SparseTensor: unsupported arguments {sorted(unsupported)}")
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("SparseTensor needs coordinates or a coordinate manager")
            if coordinates.dtype != torch.int32:
                raise ValueError("coordinates must be int32 (cast with .int(), as trainer_lighting_2d.py:151 does)")
            _lib.require_gpu(coordinates, "coordinates")
            _lib.require_gpu(features, "features")
            coordinate_manager = CoordinateManager(coordinates.device)
            uniq, _ = coordinate_manager.insert(coordinates)
            coordinate_manager.uniq = uniq
            if uniq is not None:
                features = features[uniq.long()]
            coordinate_map_key = 1
        else:
            # a manager prepared ahead of time (CoordinateManager.prepare) is handed to this stream on first use
            coordinate_manager.handover()
            if coordinates is not None and coordinate_map_key is None:
                # input tensor of a prepared manager: duplicates were dropped when the map was built, so the same
                # rows are dropped from EVERY feature matrix that enters with these coordinates (also on a second
                # forward pass over the same batch)
                if coordinate_manager.uniq is not None:
                    features = features[coordinate_manager.uniq.long()]
                coordinate_map_key = 1
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key if coordinate_map_key is not None else tensor_stride
        if features is not None and self.coordinate_map_key in coordinate_manager.maps and \
                features.shape[0] != coordinate_manager.maps[self.coordinate_map_key].n:
            raise ValueError(f"{features.shape[0]} feature rows on a coordinate map of "
                             f"{coordinate_manager.maps[self.coordinate_map_key].n} rows")
        self._F = features

    F = property(lambda self: self._F)
    C = property(lambda self: self.coordinate_manager.maps[self.coordinate_map_key].coords)
    device = property(lambda self: self._F.device)
    shape = property(lambda self: self._F.shape)
    tensor_stride = property(lambda self: [self.coordinate_map_key] * 3)

    def _like(self, feats):
        return SparseTensor(feats, coordinate_manager=self.coordinate_manager,
                            coordinate_map_key=self.coordinate_map_key)

    def _same_map(self, other):
        if other.coordinate_map_key != self.coordinate_map_key or \
                other.coordinate_manager is not self.coordinate_manager:
            raise ValueError("sparse tensors must share the coordinate map")

    def __iadd__(self, other):
        self._same_map(other)
        self._F = _AddFn.apply(self._F, other._F)
        return self

    def __add__(self, other):
        self._same_map(other)
        return self._like(_AddFn.apply(self._F, other._F))


def cat(*tensors):
    """ME.cat: feature matrices of tensors on ONE coordinate map side by side (minkunet_bev.py:337,348,359,370)"""
    for t in tensors[1:]:
        tensors[0]._same_map(t)
    out = tensors[0].F
    for t in tensors[1:]:
        out = _Cat2Fn.apply(out, t.F)
    return tensors[0]._like(out)


# ------------------------------------------------------------------ autograd functions over the C ABI
class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        call("lidog_add", ptr(a), ptr(b), a.numel(), ptr(out))
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class _Cat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        n, Ca, Cb = a.shape[0], a.shape[1], b.shape[1]
        if b.shape[0] != n:
            raise ValueError("cat: feature matrices of different lengths")
        out = torch.empty((n, Ca + Cb), dtype=torch.float32, device=a.device)
        call("lidog_cat2", ptr(a), Ca, ptr(b), Cb, n, ptr(out))
        ctx.shape = (n, Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, g):
        n, Ca, Cb = ctx.shape
        g = g.contiguous()
        ga = torch.empty((n, Ca), dtype=torch.float32, device=g.device)
        gb = torch.empty((n, Cb), dtype=torch.float32, device=g.device)
        call("lidog_split2", ptr(g), Ca, Cb, n, ptr(ga), ptr(gb))
        return ga, gb


class _ReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inplace):
        x = x.contiguous()
        y = x if inplace else torch.empty_like(x)
        call("lidog_relu_fwd", ptr(x), x.numel(), ptr(y))
        if inplace:
            ctx.mark_dirty(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(g)
        call("lidog_relu_bwd", ptr(g), ptr(y), g.numel(), ptr(dx))
        return dx, None


def _grad_out(param, shape):
    """Fresh view into the optimiser's flat gradient buffer for `param` (lidog_amd.optim.FlatParams), or None.
    A backward kernel that writes its parameter gradient there and returns the view lets autograd adopt it as
    param.grad without the `grad += new` pass (one extra kernel per parameter per step otherwise).
    The slice is handed out at most ONCE per parameter and backward pass (generation counter of FlatParams,
    advanced by zero_grad): when a parameter is used by two autograd nodes of one graph -- the reference's
    multi-source pipelines call the model twice before one backward (trainer_lighting_2d_multi.py:166-167) -- the
    second node gets None, computes into a fresh tensor, and autograd ACCUMULATES it into the view.  Before that
    second node runs, the main stream joins the weight-gradient lane: the first node's kernel may still be writing
    the slice there."""
    ref = getattr(param, "_flat_ref", None)
    if ref is None or param.grad is not None:
        return None
    buf, off, owner = ref
    if param._flat_taken == owner.generation:
        lane = wgrad_lane(buf.device) if buf.is_cuda else None
        if lane is not None:
            torch.cuda.current_stream(buf.device).wait_stream(lane.stream)
        return None
    param._flat_taken = owner.generation
    return buf[off:off + param.numel()].view(shape)


def _gemm(A, gather, B, bias, m, Cin, Cout, out, scatter, tiles=None):
    """`tiles`: (descriptors, count, pairs) of a subset of the rule book, default all of it"""
    desc, n_tiles = (tiles[0], tiles[1]) if tiles is not None else (m.tiles, m.n_tiles)
    call("lidog_sconv_gemm", ptr(A), ptr(gather), ptr(B), ptr(bias), ptr(desc[0]), ptr(desc[1]), ptr(desc[2]),
         n_tiles, Cin, Cout, ptr(out), ptr(scatter), A.shape[0])


# row of the 5^3 neighbour table that holds offset k of the 3^3 kernel (both x fastest, centred)
_SUBSET_3_OF_5 = np.array([(dz + 2) * 25 + (dy + 2) * 5 + (dx + 2) for dz in (-1, 0, 1) for dy in (-1, 0, 1)
                           for dx in (-1, 0, 1)], dtype=np.int32)
_SUBSET_MAPS = os.environ.get("LIDOG_MAP_SUBSET", "1") != "0"     # A/B switch: 0 = probe the 3^3 map as every other one
# occupancy bitmaps in front of the kernel maps' hash probes (CoordinateManager._bitmap); 0 = plain probes
_BITMAPS = os.environ.get("LIDOG_MAP_BITMAPS", "1") != "0"
_BITMAP_MAX_BYTES = int(os.environ.get("LIDOG_MAP_BITMAP_MAX_MB", "1024")) << 20
_BITMAP_MIN_ROWS = 40000


# workgroups of one weight-gradient launch (8 per CU), measured optimum on MI355X (LIDOG_WGRAD_BLOCKS: A/B runs)
_WGRAD_TARGET_BLOCKS = int(os.environ.get("LIDOG_WGRAD_BLOCKS", "2048"))
# the >= 256 x 256 layers (few pairs, 256 KB partial slots): half of that (round 5: 2 048 -> 1 024 halved their slot traffic)
_WGRAD_WIDE_BLOCKS = _WGRAD_TARGET_BLOCKS // 2


# LIDOG_WGRAD_FIT: 1 = the work items of a launch fill a whole number of rounds of the kernel's slots (see _wgrad_chunk;
# rounds = nearest to the measured optimum, 2 = rounded up); 0 = items ~ the optimum + one partial item per offset, i.e.
# usually a dozen workgroups MORE than a whole number of rounds.  Default (-1): 1 when the weight gradients run in line on
# the launch stream, 0 when they run on the second stream.  Measured (same box, alternating, round 5): in line the fitted
# launches save 0.7 ms of kernel time per step (51.16 -> 50.44 ms); on the second stream they LOSE 0.25 ms (47.48 ->
# 47.75): the loose last round of a lane kernel is when the launch stream's kernels get the chip to themselves.
_WGRAD_FIT = int(os.environ.get("LIDOG_WGRAD_FIT", "-1"))
_wgrad_slots_cache = {}


def _wgrad_slots(Cin, Cout):
    key = (Cin, Cout)
    if key not in _wgrad_slots_cache:
        _wgrad_slots_cache[key] = int(_lib.load().lidog_sconv_wgrad_slots(Cin, Cout, 0))
    return _wgrad_slots_cache[key]


def _wgrad_chunk(k_off_host, Cin, Cout):
    """pairs per weight-gradient work item for a rule book with offsets k_off_host [K + 1]"""
    P = int(k_off_host[-1]) - int(k_off_host[0])

    def tile(c):
        for t in (128, 96, 64, 32):
            if c % t == 0:
                return t
        return c
    tiles = max(1, (Cin // tile(Cin)) * (Cout // tile(Cout)))
    # measured on MI355X (scripts/sweep_wgrad.py): 8 workgroups per CU for the wide layers, 4 for the narrow ones
    # (their partial slots are cheap to compute and the final slot sum dominates)
    blocks = _WGRAD_TARGET_BLOCKS if Cin * Cout >= 128 * 128 else _WGRAD_TARGET_BLOCKS // 2
    if Cin * Cout >= 256 * 256:
        blocks = _WGRAD_WIDE_BLOCKS
    target = max(1, blocks // tiles)
    max_items = max(1, (192 << 20) // (4 * Cin * Cout))             # at most ~192 MB of partial slots
    fit = _WGRAD_FIT if _WGRAD_FIT >= 0 else (0 if _WgradLane.enabled else 1)
    slots = _wgrad_slots(Cin, Cout) if fit else 0
    if slots and blocks > slots:
        # Every offset ends in a partial item (P / chunk items are really ~K / 2 more) and the swept optimum is not a
        # multiple of the slots: 2 048 wanted on 768 slots = 2.7 rounds.  Cut so that the launch is `rounds` whole rounds.
        rounds = -(-blocks // slots) if fit == 2 else max(1, int(blocks / slots + 0.5))
        budget = min(max(1, rounds * slots // tiles), max_items)
        cnt = np.diff(np.asarray(k_off_host, dtype=np.int64))
        budget = max(budget, int(np.count_nonzero(cnt)))    # every non-empty offset costs at least one item
        chunk = max(128, (-(-P // budget) + 31) // 32 * 32)
        while int(np.sum((cnt + chunk - 1) // chunk)) > budget:
            chunk += 32
        return chunk
    chunk = max(128, -(-P // target))
    chunk = max(chunk, -(-P // max_items))
    return (chunk + 31) // 32 * 32


# launch order of the work items (csrc/hostprep.hip): 2 = items at the same relative position of their offsets together,
# in groups of 32 that land on one XCD (rounds 2-3 measured 0 = by offset and 1 = interleaved as slower)
_WGRAD_ORDER = 2
_WGRAD_GROUP = 32


def _wgrad_items_host(k_off_host, chunk):
    """(items int32 [4, n] = (k, first pair, end pair, launch order), n, item_off int32 [K+1]) as numpy arrays.
    Rows 0-2 are ordered by k (the partial slots of one offset are contiguous); row 3 says which item workgroup x
    runs: pairs are sorted by output row inside an offset, so the items at the same relative position of their
    offsets read (nearly) the same feature and gradient rows -- they are launched together, and in groups that
    land on the same XCD (workgroups go to the 8 XCDs round-robin), so that a row fetched for one offset is still
    in that XCD's L2 for the others.  (csrc/hostprep.hip)"""
    k_off = np.ascontiguousarray(k_off_host, dtype=np.int64)
    K = k_off.shape[0] - 1
    cap = int(k_off[-1] - k_off[0]) // int(chunk) + K + 1
    items = np.empty(4 * cap, dtype=np.int32)
    item_off = np.empty(K + 1, dtype=np.int32)
    n = _lib.load().lidog_wgrad_items_host(k_off.ctypes.data, K, int(chunk), _WGRAD_ORDER, _WGRAD_GROUP,
                                           items.ctypes.data, item_off.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("lidog_wgrad_items_host: capacity")
    return items[:4 * n].reshape(4, n), int(n), item_off


def _wgrad_items(m, Cin, Cout):
    """Work items of the weight gradient: the rule book cut into pair ranges of equal length that never straddle
    an offset (the centre offset of a 3^3 kernel owns one pair per voxel, corner offsets a few per cent of
    that: equal splits per offset would leave most workgroups idle behind the centre ones).
    Returns (items int32 [3, n] on the device, n, item_off int32 [K+1] on the device); cached on the map
    (CoordinateManager.prefetch fills the cache ahead of the backward pass)."""
    chunk = _wgrad_chunk(m.k_off_host, Cin, Cout)
    cache = m.__dict__.setdefault("_wgrad_items", {})
    if chunk not in cache:
        items, total, item_off = _wgrad_items_host(m.k_off_host, chunk)
        dev = m.k_off.device
        cache[chunk] = (torch.from_numpy(items).to(dev), total, torch.from_numpy(item_off).to(dev))
    return cache[chunk]


def _os_rows(m, swap, Cin, Cout):
    """the sorted rows of the map if this convolution takes the output-stationary kernel, else None"""
    if swap or isinstance(m, _IdentityMap) or Cin % 32 or Cout % 32 or _lib.load().lidog_get_sparse_core() != 1:
        return None
    return m.sorted()


class _SparseConvFn(torch.autograd.Function):
    """out = conv(x) over a rule book.  `single_out`: every output row has exactly one pair (transposed
    k2 s2) -> the GEMM scatters straight into `out`; `single_in`: every input row has exactly one pair
    (k2 s2) -> the data gradient scatters straight into `gin`."""

    @staticmethod
    def forward(ctx, x, W, bias, m, swap, single_out, single_in, stats=None, skip=False):
        """`stats`: optional StatsRequest of the BatchNorm that follows; when the output goes through the reduction
        pass, that pass also produces the fp64 sums of `out` (and, for a local BatchNorm, mean / invstd / running
        statistics), returned in the request."""
        x = x.contiguous()
        W3 = W.contiguous().view(m.K, W.shape[-2], W.shape[-1])
        K, Cin, Cout = W3.shape
        identity = isinstance(m, _IdentityMap)
        if identity:
            g_in = g_out = None
            n_out = m.n_out
        elif not swap:
            g_in, g_out, pos_o, n_out = m.pair_in, m.pair_out, m.pos_out, m.n_out
        else:  # transposed convolution: the forward map used with in/out exchanged
            g_in, g_out, pos_o, n_out = m.pair_out, m.pair_in, m.pos_in, m.n_in
        out = torch.empty((n_out, Cout), dtype=torch.float32, device=x.device)
        if identity:
            _gemm(x, None, W3, bias, m, Cin, Cout, out, None)
        elif single_out:
            _gemm(x, g_in, W3, bias, m, Cin, Cout, out, g_out)
        elif Cin == 1 and not swap and Cout in (16, 32, 64) and K * Cout * 4 <= 48 * 1024 and m.nbr is not None:
            # the stem: straight from the neighbour table, no product rows (bit-identical to the two-pass path); the
            # BatchNorm statistics of its 32-channel output are then one small pass of their own
            call("lidog_sconv_cin1", ptr(x), ptr(m.nbr), ptr(W3), ptr(bias), n_out, K, Cout, ptr(out))
        elif _os_rows(m, swap, Cin, Cout) is not None:
            # sparse symmetric 3^3 map: output-stationary kernel, no product rows (bit-identical convolution; the
            # statistics are summed per tile instead of per row block)
            perm, wmask, order = _os_rows(m, swap, Cin, Cout)
            dev = x.device
            if stats is not None:
                sums = stats.sums_out if stats.sums_out is not None else \
                    torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev)
                ws = torch.empty(_lib.load().lidog_sconv_os_stats_ws(n_out, Cout), dtype=torch.float64, device=dev)
                if stats.sync:
                    call("lidog_sconv_os_stats", ptr(x), ptr(m.nbr), n_out, K, ptr(perm), ptr(wmask), ptr(order), ptr(W3),
                         ptr(bias), Cin, Cout, ptr(out), ptr(sums), ptr(ws), float(n_out), 0.0, 0.0, None, None, None,
                         None)
                else:
                    stats.mean = torch.empty(Cout, dtype=torch.float32, device=dev)
                    stats.invstd = torch.empty(Cout, dtype=torch.float32, device=dev)
                    call("lidog_sconv_os_stats", ptr(x), ptr(m.nbr), n_out, K, ptr(perm), ptr(wmask), ptr(order), ptr(W3),
                         ptr(bias), Cin, Cout, ptr(out), ptr(sums), ptr(ws), float(n_out), stats.eps, stats.momentum,
                         ptr(stats.mean), ptr(stats.invstd), ptr(stats.running_mean), ptr(stats.running_var))
                stats.sums = sums
            else:
                call("lidog_sconv_os", ptr(x), ptr(m.nbr), n_out, K, ptr(perm), ptr(wmask), ptr(order), ptr(W3), 0,
                     ptr(bias), None, Cin, Cout, ptr(out))
        else:
            T = torch.empty((m.P, Cout), dtype=torch.float32, device=x.device)
            _gemm(x, g_in, W3, None, m, Cin, Cout, T, None)
            if Cout % 4 == 0:
                row_ptr, row_list = m.rows("in" if swap else "out")
            if stats is not None and Cout % 4 == 0 and Cout <= 1024:
                dev = x.device
                sums = stats.sums_out if stats.sums_out is not None else \
                    torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev)
                ws = torch.empty(_lib.load().lidog_sconv_reduce_stats_ws(n_out, Cout), dtype=torch.float64, device=dev)
                if stats.sync:   # the sums (and the row count behind them) still have to be all-reduced
                    call("lidog_sconv_reduce_rows_stats", ptr(T), ptr(row_ptr), ptr(row_list), n_out, Cout, ptr(bias),
                         ptr(out), ptr(sums), ptr(ws), float(n_out), 0.0, 0.0, None, None, None, None)
                else:            # local BatchNorm: mean / invstd / running statistics finalised in the same launch
                    stats.mean = torch.empty(Cout, dtype=torch.float32, device=dev)
                    stats.invstd = torch.empty(Cout, dtype=torch.float32, device=dev)
                    call("lidog_sconv_reduce_rows_stats", ptr(T), ptr(row_ptr), ptr(row_list), n_out, Cout, ptr(bias),
                         ptr(out), ptr(sums), ptr(ws), float(n_out), stats.eps, stats.momentum, ptr(stats.mean),
                         ptr(stats.invstd), ptr(stats.running_mean), ptr(stats.running_var))
                stats.sums = sums
            elif Cout % 4 == 0:
                call("lidog_sconv_reduce_rows", ptr(T), ptr(row_ptr), ptr(row_list), n_out, Cout, ptr(bias), None,
                     ptr(out))
            else:
                call("lidog_sconv_reduce", ptr(T), ptr(pos_o), n_out, K, Cout, ptr(bias), None, ptr(out))
        ctx.save_for_backward(x, W3)
        ctx.m, ctx.swap, ctx.single_in, ctx.has_bias, ctx.w_shape = m, swap, single_in, bias is not None, W.shape
        ctx.w_param, ctx.b_param = W, bias
        if skip:
            # second output = the input itself (the residual branch of a BasicBlock): its gradient comes back to
            # this node and is added by the data gradient's reduction pass instead of by an autograd add kernel
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, gout, gskip=None):
        x, W3 = ctx.saved_tensors
        m, swap = ctx.m, ctx.swap
        K, Cin, Cout = W3.shape
        gout = gout.contiguous()
        gskip = gskip.contiguous() if gskip is not None else None
        identity = isinstance(m, _IdentityMap)
        if identity:
            g_in = g_out = m.rows
            pos_i, n_in = None, m.n_in
        elif not swap:
            g_in, g_out, pos_i, n_in = m.pair_in, m.pair_out, m.pos_in, m.n_in
        else:
            g_in, g_out, pos_i, n_in = m.pair_out, m.pair_in, m.pos_out, m.n_out
        gx = gW = gb = None
        lane_on = _WgradLane.active()
        # forked behind the data gradient's GEMM (see _WgradLane), or before everything in mode 1
        behind = lane_on and _WgradLane.mode == 2

        def queue_wgrad():
            gW = _grad_out(ctx.w_param, W3.shape)
            items, n_items, item_off = _wgrad_items(m, Cin, Cout)
            slabs = _lib.load().lidog_sconv_wgrad_slabs(Cin, Cout, n_items)

            partial = torch.empty((max(slabs, 1), Cin, Cout), dtype=torch.float32, device=x.device)
            if gW is not None and lane_on:
                # launched on the lane's raw stream (no switch of torch's current stream: 2 x ~10 us of host time per
                # convolution); the scratch comes from the main stream's pool and is kept alive until the join, and
                # the lane has just been made to wait for everything queued on the main stream (fork)
                lane = _WgradLane.get(x.device)
                lane.fork(x, gout, m, partial)
                call_on(lane.raw, "lidog_sconv_wgrad", ptr(x), ptr(g_in), ptr(gout), ptr(g_out), ptr(items), n_items,
                        ptr(item_off), K, Cin, Cout, ptr(partial), ptr(gW))
            else:
                gW = gW if gW is not None else torch.empty_like(W3)
                call("lidog_sconv_wgrad", ptr(x), ptr(g_in), ptr(gout), ptr(g_out), ptr(items), n_items,
                     ptr(item_off), K, Cin, Cout, ptr(partial), ptr(gW))
            return gW.view(ctx.w_shape)

        if ctx.needs_input_grad[1] and not behind:
            gW = queue_wgrad()
        if ctx.needs_input_grad[0]:
            wp = ctx.w_param
            if getattr(wp, "_wt_version", -2) == wp._version:
                Wt = wp._wt_view                     # refreshed after the optimiser step (optim.TransposedKernels)
            else:
                Wt = torch.empty((K, Cout, Cin), dtype=torch.float32, device=x.device)
                call("lidog_transpose_kernel", ptr(W3), K, Cin, Cout, ptr(Wt))
            gx = torch.empty((n_in, Cin), dtype=torch.float32, device=x.device)
            if identity:
                _gemm(gout, None, Wt, None, m, Cout, Cin, gx, None)
            elif ctx.single_in:
                _gemm(gout, g_out, Wt, None, m, Cout, Cin, gx, g_in)
            elif _os_rows(m, swap, Cin, Cout) is not None:
                # the data gradient over the mirrored offsets of the same sorted rows (csrc/sconv_os.hip); the residual
                # branch's gradient is added in its epilogue, as the reduction pass does
                perm, wmask, order = _os_rows(m, swap, Cin, Cout)
                call("lidog_sconv_os", ptr(gout), ptr(m.nbr), n_in, K, ptr(perm), ptr(wmask), ptr(order), ptr(Wt), 1,
                     None, ptr(gskip), Cout, Cin, ptr(gx))
                gskip = None
                if ctx.needs_input_grad[1] and behind:
                    gW = queue_wgrad()
            else:
                T = torch.empty((m.P, Cin), dtype=torch.float32, device=x.device)
                _gemm(gout, g_out, Wt, None, m, Cout, Cin, T, None)
                if ctx.needs_input_grad[1] and behind:
                    gW = queue_wgrad()
                add = gskip if (gskip is not None and Cin % 4 == 0) else None
                if Cin % 4 == 0:
                    row_ptr, row_list = m.rows("out" if swap else "in")
                    call("lidog_sconv_reduce_rows", ptr(T), ptr(row_ptr), ptr(row_list), n_in, Cin, None, ptr(add),
                         ptr(gx))
                else:
                    call("lidog_sconv_reduce", ptr(T), ptr(pos_i), n_in, K, Cin, None, None, ptr(gx))
                if add is not None:
                    gskip = None
        if ctx.needs_input_grad[1] and gW is None:
            gW = queue_wgrad()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _grad_out(ctx.b_param, (1, Cout))
            gb = gb if gb is not None else torch.empty((1, Cout), dtype=torch.float32, device=x.device)
            ws = torch.empty(_lib.load().lidog_colsum_ws(Cout), dtype=torch.float64, device=x.device)
            call("lidog_colsum", ptr(gout), gout.shape[0], Cout, ptr(gb), ptr(ws))
        if gskip is not None:   # paths without a reduction pass (1x1, k2 s2) or no data gradient asked for
            gx = gskip if gx is None else gx + gskip
        return gx, gW, gb, None, None, None, None, None, None


def _bn_ws(C, hw, dev, images=1):
    """scratch of the BatchNorm reductions (per-workgroup partials; NCHW input: per image)"""
    n = _lib.load().lidog_bn_reduce_ws(C, hw) * (images if hw > 1 else 1)
    return torch.empty(n, dtype=torch.float64, device=dev) if n else None


class StatsRequest:
    """What a convolution needs to know to produce the statistics of the BatchNorm that follows it in the
    epilogue of its reduction pass (conv_bn), and where it leaves them."""
    __slots__ = ("eps", "momentum", "running_mean", "running_var", "sync", "sums", "mean", "invstd", "sums_out",
                 "presynced")

    def __init__(self, bn, sync, momentum):
        self.eps = float(bn.eps)
        self.momentum = float(momentum)
        self.running_mean, self.running_var = bn.running_mean, bn.running_var
        self.sync = sync
        self.sums = self.mean = self.invstd = None
        self.sums_out = None      # where the (sum x, sum x^2, rows) vector goes (a slice of a joint all-reduce buffer)
        self.presynced = False    # the sums have been all-reduced already (together with another layer's)


class _BatchNormFn(torch.autograd.Function):
    """BatchNorm over rows ([n,C], hw=1) or NCHW images (hw=H*W), optional fused residual add and ReLU.
    `group`: torch.distributed process group for SyncBatchNorm statistics (None = local).
    `pre`: optional (sums, mean, invstd) from the epilogue of the convolution that made x (saves one pass over
    x): fp64 sums [2C+1] = (sum x, sum x^2, rows); mean / invstd None when they still have to be derived
    (SyncBatchNorm: after the all-reduce).
    SyncBatchNorm costs ONE collective per direction and no extra kernel: the reductions leave the local row
    count behind their sums, the all-reduce turns both into global figures, the consumers read the count from
    the device."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, hw, relu, residual, group,
                pre):
        x = x.contiguous()
        if hw == 1:
            n, C = x.shape
        else:
            n, C = x.shape[0], x.shape[1]
        rows = float(n * hw)
        count = rows
        dev = x.device
        if training:
            sums, mean, invstd, presynced = pre if pre is not None else (None, None, None, False)
            sync = group is not None
            if sums is None:
                sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
                if sync:
                    call("lidog_bn_stats", ptr(x), n, C, hw, ptr(sums), ptr(_bn_ws(C, hw, dev, n)), rows, 0.0, 0.0,
                         None, None, None, None)
                else:
                    mean = torch.empty(C, dtype=torch.float32, device=dev)
                    invstd = torch.empty(C, dtype=torch.float32, device=dev)
                    call("lidog_bn_stats", ptr(x), n, C, hw, ptr(sums), ptr(_bn_ws(C, hw, dev, n)), rows, float(eps),
                         float(momentum), ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var))
            if sync:
                if not presynced:
                    from .comm import transport
                    transport(group).allreduce_f64(sums)   # (sum x, sum x^2, rows) in ONE message
                count = -1.0                              # consumers read the global count from sums[2C]
            if mean is None:
                mean = torch.empty(C, dtype=torch.float32, device=dev)
                invstd = torch.empty(C, dtype=torch.float32, device=dev)
                call("lidog_bn_finalize", ptr(sums), count, C, float(eps), float(momentum), ptr(mean), ptr(invstd),
                     ptr(running_mean), ptr(running_var))
        else:
            mean = running_mean
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            call("lidog_bn_eval_invstd", ptr(running_var), float(eps), C, ptr(invstd))
        y = torch.empty_like(x)
        res = residual.contiguous() if residual is not None else None
        call("lidog_bn_apply", ptr(x), n, C, hw, ptr(mean), ptr(invstd), ptr(weight), ptr(bias), ptr(res),
             1 if relu else 0, ptr(y))
        # BatchNorm + ReLU without residual on a [rows, C] matrix: backward recomputes the ReLU mask from x (the forward
        # pass's own expression, same bits) instead of reading y -- one tensor less in both backward kernels
        mask_from_x = relu and residual is None and hw == 1 and C % 4 == 0
        ctx.save_for_backward(x, weight, mean, invstd, y if (relu and not mask_from_x) else None,
                              bias if mask_from_x else None)
        ctx.cfg = (n, C, hw, rows, training, residual is not None, group)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, invstd, y, mask_b = ctx.saved_tensors
        mask_w = weight if mask_b is not None else None
        n, C, hw, rows, training, has_res, group = ctx.cfg
        dy = dy.contiguous()
        dev = x.device
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        # parameter gradients are the LOCAL sums (DDP averages them afterwards), written by the last kernel of the
        # reduction straight into the optimiser's flat gradient buffer when there is one
        dw = _grad_out(ctx.params[0], (C,))
        db = _grad_out(ctx.params[1], (C,))
        dw = dw if dw is not None else torch.empty(C, dtype=torch.float32, device=dev)
        db = db if db is not None else torch.empty(C, dtype=torch.float32, device=dev)
        call("lidog_bn_bwd_reduce", ptr(dy), ptr(x), ptr(y), n, C, hw, ptr(mean), ptr(invstd), ptr(sums),
             ptr(_bn_ws(C, hw, dev, n)), rows, ptr(dw), ptr(db), ptr(mask_w), ptr(mask_b))
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        count = rows
        if not training:
            sums.zero_()  # running statistics are constants: dx = dy' * w * invstd
        elif group is not None:
            from .comm import transport
            transport(group).allreduce_f64(sums)   # (sum dy', sum dy' xhat, rows)
            count = -1.0
        call("lidog_bn_bwd_apply", ptr(dy), ptr(x), ptr(y), n, C, hw, ptr(mean), ptr(invstd), ptr(weight),
             ptr(sums), count, ptr(dx), ptr(dres), None, None, ptr(mask_b))
        return dx, dw, db, None, None, None, None, None, None, None, dres, None, None


def _flush_batch_counter(bn, *_):
    n = getattr(bn, "_nbt_pending", 0)
    if n:
        bn._nbt_pending = 0
        bn.num_batches_tracked.add_(n)


def _count_batch(bn):
    """num_batches_tracked += 1 counted on the host (62 one-element kernel launches per step otherwise); the
    buffer is brought up to date whenever a state_dict is taken."""
    bn._nbt_pending = getattr(bn, "_nbt_pending", 0) + 1
    if not getattr(bn, "_nbt_hooked", False):
        bn._nbt_hooked = True
        bn.register_state_dict_pre_hook(_flush_batch_counter)
        bn.register_load_state_dict_pre_hook(lambda module, *_: setattr(module, "_nbt_pending", 0))


def _training_momentum(bn):
    """running-statistics update factor of ONE training-mode forward pass, counting the batch as torch does:
    bn.momentum, or 1 / num_batches_tracked (cumulative moving average) when bn.momentum is None"""
    if not (bn.track_running_stats and bn.num_batches_tracked is not None):
        return 0.0 if bn.momentum is None else float(bn.momentum)
    if bn.momentum is None:   # the factor depends on the counter: read it (one host synchronisation; no LiDOG config
        _flush_batch_counter(bn)                       # uses momentum=None, MinkowskiBatchNorm defaults to 0.1)
        count = int(bn.num_batches_tracked.item()) + 1
        bn.num_batches_tracked.add_(1)
        return 1.0 / count
    _count_batch(bn)
    return float(bn.momentum)


def batch_norm(x, bn, hw=1, relu=False, residual=None, group=None, stats=None):
    """functional entry used by the modules below and by lidog_amd.bev; `stats`: the StatsRequest a convolution
    has filled (conv_bn; the batch is then already counted), or None"""
    training = bn.training or not bn.track_running_stats
    if stats is not None:
        momentum = stats.momentum
    else:
        momentum = _training_momentum(bn) if training else 0.0
    pre = (stats.sums, stats.mean, stats.invstd, stats.presynced) \
        if (training and stats is not None and stats.sums is not None) else None
    return _BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps,
                              hw, relu, residual, group, pre)


# ------------------------------------------------------------------ modules
class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None, **_unused):
        super().__init__()
        if dimension != 3:
            raise ValueError("lidog_amd implements the 3-D operators of the LiDOG hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = int(kernel_size), int(stride), int(dilation)
        self.dimension = dimension
        self.kernel_volume = self.kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else \
            (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.kernel._lidog_sparse_kernel = True   # lidog_amd.optim.TransposedKernels keeps a [K, Cout, Cin] copy
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _resolve(self, x):
        """(kernel map, output map key, swap, single_out, single_in) of this convolution on x's coordinate map; records
        the use in the manager's trace (CoordinateManager.prepare builds the next batch's maps from it)"""
        cm, s_in = x.coordinate_manager, x.coordinate_map_key
        if self.kernel_volume == 1 and self.stride == 1:
            m, s_out, swap, single_out, single_in = cm.identity_map(x.F.shape[0]), s_in, False, True, True
            cm.trace.append((("identity", s_in), self.in_channels, self.out_channels))
        elif not self.transposed:
            s_out = s_in * self.stride
            m = cm.kernel_map(s_in, s_out, self.kernel_size, self.dilation)
            cm.trace.append(((s_in, s_out, self.kernel_size, self.dilation), self.in_channels, self.out_channels))
            swap, single_out = False, False
            single_in = self.stride == self.kernel_size and self.stride > 1  # non-overlapping windows
        else:
            if s_in % self.stride != 0 or (s_in // self.stride) not in cm.maps:
                raise ValueError("transposed convolution must land on an existing finer coordinate map")
            s_out = s_in // self.stride
            m = cm.kernel_map(s_out, s_in, self.kernel_size, self.dilation)
            cm.trace.append(((s_out, s_in, self.kernel_size, self.dilation), self.in_channels, self.out_channels))
            swap, single_in = True, False
            single_out = self.stride == self.kernel_size and self.stride > 1
        return m, s_out, swap, single_out, single_in

    def forward(self, x, stats=None, skip=False):
        """`skip=True` returns (conv(x), x'): x' is x again, but routed through this convolution's autograd node
        so that the gradient of a residual branch taken from x' is added inside the data gradient's reduction"""
        cm, s_in = x.coordinate_manager, x.coordinate_map_key
        m, s_out, swap, single_out, single_in = self._resolve(x)
        out = _SparseConvFn.apply(x.F, self.kernel, self.bias, m, swap, single_out, single_in, stats, skip)
        if skip:
            return (SparseTensor(out[0], coordinate_manager=cm, coordinate_map_key=s_out),
                    SparseTensor(out[1], coordinate_manager=cm, coordinate_map_key=s_in))
        return SparseTensor(out, coordinate_manager=cm, coordinate_map_key=s_out)

    def forward_eval_bn(self, x, bn, relu, residual):
        """Validation path, no autograd: convolution whose reduction pass applies the evaluation-mode BatchNorm
        (+ residual + ReLU) in its epilogue (lidog_sconv_reduce_rows_bn) -- no separate BatchNorm kernel, no round
        trip of the convolution output.  Returns None when this convolution does not go through the reduction
        pass (1x1, transposed k2 s2, the 5^3 stem, channel counts not a multiple of 4): the caller falls back."""
        if self.kernel_volume == 1 or self.transposed or self.in_channels == 1 or self.out_channels % 4:
            return None
        cm = x.coordinate_manager
        m, s_out, swap, single_out, single_in = self._resolve(x)
        xf = x.F.contiguous()
        W3 = self.kernel.detach().contiguous().view(m.K, self.in_channels, self.out_channels)
        Cin, Cout, dev = self.in_channels, self.out_channels, xf.device
        invstd = torch.empty(Cout, dtype=torch.float32, device=dev)
        call("lidog_bn_eval_invstd", ptr(bn.running_var), float(bn.eps), Cout, ptr(invstd))
        out = torch.empty((m.n_out, Cout), dtype=torch.float32, device=dev)
        res = residual.F.contiguous() if residual is not None else None
        srt = _os_rows(m, swap, Cin, Cout)
        if srt is not None and not single_out:      # sorted rows: output-stationary kernel, same epilogue (same bits)
            call("lidog_sconv_os_bn", ptr(xf), ptr(m.nbr), m.n_out, m.K, ptr(srt[0]), ptr(srt[1]), ptr(srt[2]), ptr(W3),
                 ptr(self.bias.detach()) if self.bias is not None else None, Cin, Cout, ptr(bn.running_mean), ptr(invstd),
                 ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(res), 1 if relu else 0, ptr(out))
            return SparseTensor(out, coordinate_manager=cm, coordinate_map_key=s_out)
        T = torch.empty((m.P, Cout), dtype=torch.float32, device=dev)
        _gemm(xf, m.pair_in, W3, None, m, Cin, Cout, T, None)
        row_ptr, row_list = m.rows("out")
        call("lidog_sconv_reduce_rows_bn", ptr(T), ptr(row_ptr), ptr(row_list), m.n_out, Cout,
             ptr(self.bias.detach()) if self.bias is not None else None, ptr(bn.running_mean), ptr(invstd),
             ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(res), 1 if relu else 0, ptr(out))
        return SparseTensor(out, coordinate_manager=cm, coordinate_map_key=s_out)


class MinkowskiConvolution(_ConvBase):
    transposed = False


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    _group = None

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x, relu=False, residual=None, stats=None):
        res = residual.F if residual is not None else None
        return x._like(batch_norm(x.F, self.bn, 1, relu, res, self._sync_group(), stats))

    def _sync_group(self):
        return None


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """Statistics all-reduced over the process group (RCCL); parameter names unchanged so checkpoints
    interchange with the unsynchronised model (train_lidog.py:228, eval_target.py:153)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group

    single_rank = False   # test hook: run the collectives even in a one-rank process group

    def _sync_group(self):
        import torch.distributed as dist
        if not (self.training and dist.is_available() and dist.is_initialized()) or \
                (dist.get_world_size() == 1 and not MinkowskiSyncBatchNorm.single_rank):
            return None
        return self.process_group if self.process_group is not None else dist.group.WORLD

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            out = cls(module.bn.num_features, module.bn.eps, module.bn.momentum, module.bn.affine,
                      module.bn.track_running_stats, process_group)
            out.bn = module.bn
            out.training = module.training
        for name, child in module.named_children():
            if out is module or name != "bn":
                out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        return out


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        f = x.F
        inplace = self.inplace and not (f.requires_grad and f.is_leaf)
        return x._like(_ReLUFn.apply(f, inplace))


def conv_bn(conv, bn_module, x, relu=False, residual=None, skip=False):
    """convolution + BatchNorm (+ residual add + ReLU): the BN statistics come out of the convolution's
    reduction pass, the affine/add/ReLU is one fused elementwise pass.  `skip`: also return x routed through the
    convolution's autograd node (MinkowskiConvolution.forward)."""
    bn = bn_module.bn
    req = None
    if bn.training or not bn.track_running_stats:
        req = StatsRequest(bn, bn_module._sync_group() is not None, _training_momentum(bn))
    elif not skip and not torch.is_grad_enabled() and bn.affine:
        # validation path under no_grad: BatchNorm applied in the reduction pass's epilogue where there is one
        y = conv.forward_eval_bn(x, bn, relu, residual)
        if y is not None:
            return y
    if skip:
        y, x_alias = conv(x, stats=req, skip=True)
        return bn_module(y, relu=relu, residual=residual, stats=req), x_alias
    y = conv(x, stats=req)
    return bn_module(y, relu=relu, residual=residual, stats=req)


def bn_relu(bn_module, x):
    """BatchNorm + ReLU as one fused pass (used by lidog_amd.minkunet when the backend offers it)"""
    return bn_module(x, relu=True)


class MinkowskiDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.drop = nn.Dropout(p, inplace)

    def forward(self, x):
        return x._like(self.drop(x.F))


# ------------------------------------------------------------------ utils
def _fans(t):
    if t.dim() == 2:
        return t.size(0), t.size(1)
    return t.size(1) * t.size(0), t.size(2) * t.size(0)


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    fan_in, fan_out = _fans(tensor)
    std = nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan_in if mode == "fan_in" else fan_out)
    with torch.no_grad():
        return tensor.normal_(0, std)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        out.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=c.dtype), c], dim=1))
    return torch.cat(out, dim=0).to(dtype)


class SparseCollation:
    """ME.utils.SparseCollation (utils/collation/collation.py:309): batch index prepended as column 0."""

    def __init__(self, limit_numpoints=-1, dtype=torch.int32, device=None):
        self.dtype, self.device = dtype, device

    def __call__(self, list_data):
        coords, feats, labels = list(zip(*list_data))
        return (batched_coordinates(coords, dtype=self.dtype),
                torch.cat([torch.as_tensor(f) for f in feats], dim=0),
                torch.cat([torch.as_tensor(l) for l in labels], dim=0))


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device="cuda"):
    """ME.utils.sparse_quantize with MinkowskiEngine 0.5.4's signature and return convention, as the reference
    calls it: utils/datasets/semantickitti_bev.py:232-238 (5 results), utils/datasets/mix3D.py:67-72 (4),
    utils/models/minkunet_bev.py:279-284 (3, vector quantization_size).  numpy arrays in -> numpy arrays out, torch
    tensors in -> torch tensors on the input's device out; the work (floor-divide, hash unique, label vote) runs in
    the HIP kernels of lidog_amd.data.sparse_quantize -- there is no CPU implementation, so a DataLoader worker that
    calls this needs the GPU (start method 'spawn'), see INTEGRATION.md."""
    from . import data as _data
    is_np = isinstance(coordinates, np.ndarray)

    def to_dev(a):
        if a is None:
            return None
        return (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(device)

    def back(t, like):
        if is_np:
            return t.cpu().numpy()
        return t.to(like.device)

    pts = to_dev(coordinates).float()
    q = 1 if quantization_size is None else quantization_size
    if torch.is_tensor(q):
        q = q.cpu().numpy()
    res = _data.sparse_quantize(pts, None, labels=to_dev(labels), ignore_label=ignore_label, quantization_size=q,
                                return_index=True, return_inverse=True)
    if labels is not None:
        coords, vlab, index, inverse = res
    else:
        (coords, index, inverse), vlab = res, None
    if return_maps_only:
        out = [back(index, coordinates)] + ([back(inverse, coordinates)] if return_inverse else [])
        return out[0] if len(out) == 1 else tuple(out)
    out = [back(coords, coordinates)]
    if features is not None:
        idx_host = index.cpu().numpy() if isinstance(features, np.ndarray) else index.to(features.device)
        out.append(features[idx_host])
    if labels is not None:
        out.append(back(vlab, labels))
    if return_index:
        out.append(back(index, coordinates))
    if return_inverse:
        out.append(back(inverse, coordinates))
    return out[0] if len(out) == 1 else tuple(out)


utils = types.ModuleType(__name__ + ".utils")
utils.sparse_quantize = sparse_quantize
utils.kaiming_normal_ = kaiming_normal_
utils.SparseCollation = SparseCollation
utils.batched_coordinates = batched_coordinates


# ------------------------------------------------------------------ modules.resnet_block
class BasicBlock(nn.Module):
    """conv3-BN-ReLU-conv3-BN-(+residual)-ReLU; BN+ReLU and BN+add+ReLU run as single fused kernels."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        group = self.norm1._sync_group() if self.norm1.bn.training else None
        if group is not None and isinstance(self.downsample, nn.Sequential) and len(self.downsample) == 2 and \
                isinstance(self.downsample[1], MinkowskiSyncBatchNorm) and self.downsample[1].bn.training:
            return self._forward_joint_sync(x, group)
        if x.F.requires_grad:
            # the residual branch takes x from conv1's autograd node, so the two gradients of x (conv1's data
            # gradient and the residual's / the downsample convolution's) are summed by conv1's reduction pass, not
            # by a separate add kernel
            out, x = conv_bn(self.conv1, self.norm1, x, relu=True, skip=True)
        else:
            out = conv_bn(self.conv1, self.norm1, x, relu=True)
        residual = x if self.downsample is None else self.downsample(x)
        return conv_bn(self.conv2, self.norm2, out, relu=True, residual=residual)

    def _forward_joint_sync(self, x, group):
        """First block of a layer under SyncBatchNorm: conv1 and the 1x1 downsample convolution both read x and are
        independent, so the statistics of their two BatchNorms travel in ONE all-reduce (7 collectives fewer per
        forward pass; a statistics all-reduce is pure latency on the dependent chain)."""
        from .comm import transport
        bn1, (convd, bnd) = self.norm1, self.downsample
        Ca, Cd = bn1.bn.num_features, bnd.bn.num_features
        joint = torch.empty(2 * Ca + 1 + 2 * Cd + 1, dtype=torch.float64, device=x.F.device)
        req_a = StatsRequest(bn1.bn, True, _training_momentum(bn1.bn))
        req_d = StatsRequest(bnd.bn, True, _training_momentum(bnd.bn))
        req_a.sums_out, sums_d = joint[:2 * Ca + 1], joint[2 * Ca + 1:]
        if x.F.requires_grad:
            y1, x = self.conv1(x, stats=req_a, skip=True)
        else:
            y1 = self.conv1(x, stats=req_a)
        yd = convd(x)
        if req_a.sums is None:     # conv1 did not go through the reduction pass with fused statistics
            call("lidog_bn_stats", ptr(y1.F), y1.F.shape[0], Ca, 1, ptr(req_a.sums_out), ptr(_bn_ws(Ca, 1, y1.F.device)),
                 float(y1.F.shape[0]), 0.0, 0.0, None, None, None, None)
            req_a.sums = req_a.sums_out
        call("lidog_bn_stats", ptr(yd.F), yd.F.shape[0], Cd, 1, ptr(sums_d), ptr(_bn_ws(Cd, 1, yd.F.device)),
             float(yd.F.shape[0]), 0.0, 0.0, None, None, None, None)
        req_d.sums = sums_d
        transport(group).allreduce_f64(joint)
        req_a.presynced = req_d.presynced = True
        out = bn1(y1, relu=True, stats=req_a)
        residual = bnd(yd, stats=req_d)
        return conv_bn(self.conv2, self.norm2, out, relu=True, residual=residual)


class Bottleneck(nn.Module):
    """ME's Bottleneck (1x1 -> 3^3 -> 1x1 x expansion, residual; imported by utils/models/minkunet_bev.py:4 and
    selectable as `BLOCK` of a MinkUNet variant, never instantiated by the LiDOG configurations): the same fused
    conv + BatchNorm (+ residual) + ReLU kernels as BasicBlock."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = conv_bn(self.conv1, self.norm1, x, relu=True)
        out = conv_bn(self.conv2, self.norm2, out, relu=True)
        residual = x if self.downsample is None else self.downsample(x)
        return conv_bn(self.conv3, self.norm3, out, relu=True, residual=residual)


modules = types.ModuleType(__name__ + ".modules")
resnet_block = types.ModuleType(__name__ + ".modules.resnet_block")
resnet_block.BasicBlock = BasicBlock
resnet_block.Bottleneck = Bottleneck
modules.resnet_block = resnet_block


def install_as_minkowski_engine():
    me = sys.modules[__name__]
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.utils"] = utils
    sys.modules["MinkowskiEngine.modules"] = modules
    sys.modules["MinkowskiEngine.modules.resnet_block"] = resnet_block
    return me
