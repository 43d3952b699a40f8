"""CPU oracle with the MinkowskiEngine v0.5.4 python surface LiDOG uses.

TEST INFRASTRUCTURE ONLY (see oracle/me_oracle.c header): imported by tests/,
tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by lidog_amd/.

It exists so that (1) the reference's own model file
(/root/reference/utils/models/minkunet_bev.py) can be imported in the build
container with ``sys.modules['MinkowskiEngine'] = oracle.me_cpu`` to generate
golden vectors, (2) the HIP path can be checked op by op, and (3) the CPU
baseline can be timed with ME's CPU algorithm (per-offset gather -> GEMM ->
scatter-add; ``set_mode('blas')``).

Names provided follow SURVEY.md section 8(b): SparseTensor, MinkowskiConvolution,
MinkowskiConvolutionTranspose, MinkowskiBatchNorm, MinkowskiSyncBatchNorm,
MinkowskiReLU, MinkowskiDropout, cat, utils.{kaiming_normal_, sparse_quantize,
SparseCollation, batched_coordinates}, modules.resnet_block.{BasicBlock,
Bottleneck}.
"""
import math
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from ._lib import lib

_MODE = "exact"  # "exact": C fmaf-chain kernels; "blas": torch mm per offset (ME CPU algorithm, fast)


def set_mode(mode):
    global _MODE
    assert mode in ("exact", "blas")
    _MODE = mode


def _ptr(t):
    return t.data_ptr() if t is not None else None


def kernel_offsets(kernel_size, tensor_stride, dilation=1):
    """Offsets of a hyper-cubic 3-D kernel, index runs x fastest, then y, then z.
    Odd sizes are centred, even sizes start at 0."""
    k = int(kernel_size)
    if k % 2 == 1:
        r = [(-(k // 2) + i) * tensor_stride * dilation for i in range(k)]
    else:
        r = [i * tensor_stride * dilation for i in range(k)]
    offs = [(x, y, z) for z in r for y in r for x in r]
    return torch.tensor(offs, dtype=torch.int32)


class CoordinateManager:
    """Coordinate maps keyed by tensor stride; kernel maps cached per
    (in stride, out stride, kernel size, dilation)."""

    def __init__(self):
        self.maps = {}
        self.kmaps = {}

    def insert(self, coords):
        coords = coords.contiguous().to(torch.int32)
        n = coords.shape[0]
        uniq = torch.empty(n, dtype=torch.int32)
        inv = torch.empty(n, dtype=torch.int32)
        m = lib().orc_unique_first(_ptr(coords), n, _ptr(uniq), _ptr(inv))
        uniq = uniq[:m]
        self.maps[1] = coords[uniq.long()].contiguous() if m != n else coords
        return uniq, inv

    def stride(self, s_in, s_out):
        if s_out not in self.maps:
            c = self.maps[s_in]
            out = torch.empty_like(c)
            p2c = torch.empty(c.shape[0], dtype=torch.int32)
            m = lib().orc_stride(_ptr(c), c.shape[0], s_out, _ptr(out), _ptr(p2c))
            self.maps[s_out] = out[:m].contiguous()
        return self.maps[s_out]

    def kernel_map(self, s_in, s_out, kernel_size, dilation=1):
        """pairs (k_off[K+1] int64, pair_in, pair_out) of the map in(s_in) -> out(s_out)."""
        key = (s_in, s_out, kernel_size, dilation)
        if key not in self.kmaps:
            cin = self.maps[s_in]
            cout = self.stride(s_in, s_out) if s_out != s_in else cin
            offs = kernel_offsets(kernel_size, s_in, dilation)
            K = offs.shape[0]
            nbr = torch.empty((cout.shape[0], K), dtype=torch.int32)
            P = lib().orc_kernel_map(_ptr(cin), cin.shape[0], _ptr(cout), cout.shape[0], _ptr(offs), K, _ptr(nbr))
            k_off = torch.empty(K + 1, dtype=torch.int64)
            pin = torch.empty(P, dtype=torch.int32)
            pout = torch.empty(P, dtype=torch.int32)
            lib().orc_pairs_from_nbr(_ptr(nbr), cout.shape[0], K, _ptr(k_off), _ptr(pin), _ptr(pout))
            self.kmaps[key] = (k_off, pin, pout, nbr)
        return self.kmaps[key]


class SparseTensor:
    def __init__(self, features=None, coordinates=None, coordinate_manager=None, coordinate_map_key=None,
                 tensor_stride=1, **_unused):
        if coordinate_manager is None:
            assert coordinates is not None
            assert coordinates.dtype == torch.int32, "coordinates must be int32 (ME asserts this)"
            coordinate_manager = CoordinateManager()
            uniq, inv = coordinate_manager.insert(coordinates.cpu())
            if uniq.shape[0] != coordinates.shape[0]:
                features = features[uniq.long()]
            coordinate_map_key = 1
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key if coordinate_map_key is not None else tensor_stride
        self._F = features

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.maps[self.coordinate_map_key]

    @property
    def device(self):
        return self._F.device

    @property
    def tensor_stride(self):
        return [self.coordinate_map_key] * 3

    @property
    def shape(self):
        return self._F.shape

    def __iadd__(self, other):
        assert other.coordinate_map_key == self.coordinate_map_key
        self._F = self._F + other._F
        return self

    def __add__(self, other):
        assert other.coordinate_map_key == self.coordinate_map_key
        return SparseTensor(self._F + other._F, coordinate_manager=self.coordinate_manager,
                            coordinate_map_key=self.coordinate_map_key)


def cat(*tensors):
    key = tensors[0].coordinate_map_key
    for t in tensors:
        if t.coordinate_map_key != key or t.coordinate_manager is not tensors[0].coordinate_manager:
            raise ValueError("cat: sparse tensors must share the coordinate map")
    return SparseTensor(torch.cat([t.F for t in tensors], dim=1), coordinate_manager=tensors[0].coordinate_manager,
                        coordinate_map_key=key)


def _gather(src, idx):
    idx = idx.contiguous()
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=torch.float32)
    lib().orc_gather_rows(_ptr(src), _ptr(idx), idx.shape[0], src.shape[1], _ptr(out))
    return out


def _scatter_add(dst, idx, rows):
    idx, rows = idx.contiguous(), rows.contiguous()
    lib().orc_scatter_add_rows(_ptr(rows), _ptr(idx), idx.shape[0], dst.shape[1], _ptr(dst))
    return dst


class _SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, k_off, pin, pout, n_out):
        K, Cin, Cout = W.shape
        x = x.contiguous()
        W = W.contiguous()
        out = torch.zeros((n_out, Cout), dtype=x.dtype)
        if x.dtype != torch.float32:
            # float64 ground truth for the gradient tests (tests/golden/make_golden.py g5): same algorithm in torch ops
            for k in range(K):
                a, b = int(k_off[k]), int(k_off[k + 1])
                if b > a:
                    out.index_add_(0, pout[a:b].long(), x[pin[a:b].long()] @ W[k])
        elif _MODE == "exact":
            lib().orc_conv_fwd(_ptr(x), _ptr(W), _ptr(k_off), _ptr(pin), _ptr(pout), K, Cin, Cout, _ptr(out))
        else:  # ME's CPU algorithm: per offset gather rows -> BLAS GEMM -> scatter-add rows (OpenMP helpers)
            for k in range(K):
                a, b = int(k_off[k]), int(k_off[k + 1])
                if b > a:
                    out = _scatter_add(out, pout[a:b], _gather(x, pin[a:b]) @ W[k])
        ctx.save_for_backward(x, W, k_off, pin, pout)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, W, k_off, pin, pout = ctx.saved_tensors
        K, Cin, Cout = W.shape
        gout = gout.contiguous()
        gin = torch.zeros_like(x)
        gW = torch.zeros_like(W)
        if x.dtype != torch.float32:
            for k in range(K):
                a, b = int(k_off[k]), int(k_off[k + 1])
                if b > a:
                    g = gout[pout[a:b].long()]
                    gin.index_add_(0, pin[a:b].long(), g @ W[k].t())
                    gW[k] = x[pin[a:b].long()].t() @ g
        elif _MODE == "exact":
            lib().orc_conv_bwd_data(_ptr(gout), _ptr(W), _ptr(k_off), _ptr(pin), _ptr(pout), K, Cin, Cout, _ptr(gin))
            lib().orc_conv_bwd_weight(_ptr(x), _ptr(gout), _ptr(k_off), _ptr(pin), _ptr(pout), K, Cin, Cout, _ptr(gW))
        else:
            for k in range(K):
                a, b = int(k_off[k]), int(k_off[k + 1])
                if b > a:
                    g = _gather(gout, pout[a:b])
                    xi = _gather(x, pin[a:b])
                    gin = _scatter_add(gin, pin[a:b], g @ W[k].t())
                    gW[k] = xi.t() @ g
        return gin, gW, None, None, None, None


class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None, **_unused):
        super().__init__()
        assert dimension == 3, "the LiDOG hot path is 3-D"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = int(kernel_size), int(stride), int(dilation)
        self.dimension = dimension
        self.kernel_volume = self.kernel_size ** 3
        if self.kernel_volume > 1:
            self.kernel = nn.Parameter(torch.empty(self.kernel_volume, in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, x):
        cm = x.coordinate_manager
        s_in = x.coordinate_map_key
        if self.kernel_volume == 1 and self.stride == 1:
            s_out = s_in
            if _MODE == "exact":  # same fmaf chain as every other kernel size (K = 1, in row == out row)
                n = x.F.shape[0]
                rows = torch.arange(n, dtype=torch.int32)
                out = _SparseConvFn.apply(x.F, self.kernel.view(1, self.in_channels, self.out_channels),
                                          torch.tensor([0, n], dtype=torch.int64), rows, rows, n)
            else:
                out = x.F @ self.kernel
        elif not self.transposed:
            s_out = s_in * self.stride
            k_off, pin, pout, _ = cm.kernel_map(s_in, s_out, self.kernel_size, self.dilation)
            n_out = cm.maps[s_out].shape[0]
            out = _SparseConvFn.apply(x.F, self.kernel.view(self.kernel_volume, self.in_channels, self.out_channels),
                                      k_off, pin, pout, n_out)
        else:
            assert s_in % self.stride == 0
            s_out = s_in // self.stride
            assert s_out in cm.maps, "transposed conv must land on an existing finer map"
            # forward map fine(s_out) -> coarse(s_in), used with in/out swapped
            k_off, pfine, pcoarse, _ = cm.kernel_map(s_out, s_in, self.kernel_size, self.dilation)
            n_out = cm.maps[s_out].shape[0]
            out = _SparseConvFn.apply(x.F, self.kernel, k_off, pcoarse, pfine, n_out)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, coordinate_manager=cm, coordinate_map_key=s_out)


class MinkowskiConvolution(_ConvBase):
    transposed = False


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return SparseTensor(self.bn(x.F), coordinate_manager=x.coordinate_manager,
                            coordinate_map_key=x.coordinate_map_key)


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        nn.Module.__init__(self)
        self.bn = nn.SyncBatchNorm(num_features, eps=eps, momentum=momentum, affine=affine,
                                   track_running_stats=track_running_stats, process_group=process_group)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            out = cls(module.bn.num_features, module.bn.eps, module.bn.momentum, module.bn.affine,
                      module.bn.track_running_stats, process_group)
            if module.bn.affine:
                out.bn.weight, out.bn.bias = module.bn.weight, module.bn.bias
            out.bn.running_mean, out.bn.running_var = module.bn.running_mean, module.bn.running_var
            out.bn.num_batches_tracked = module.bn.num_batches_tracked
        for name, child in module.named_children():
            out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        return out


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return SparseTensor(torch.relu(x.F), coordinate_manager=x.coordinate_manager,
                            coordinate_map_key=x.coordinate_map_key)


class MinkowskiDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.drop = nn.Dropout(p, inplace)

    def forward(self, x):
        return SparseTensor(self.drop(x.F), coordinate_manager=x.coordinate_manager,
                            coordinate_map_key=x.coordinate_map_key)


# ---------------------------------------------------------------- utils
def _fans(t):
    if t.dim() == 2:
        return t.size(0), t.size(1)  # [Cin, Cout]
    return t.size(1) * t.size(0), t.size(2) * t.size(0)  # [K, Cin, Cout]


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    fan_in, fan_out = _fans(tensor)
    fan = fan_in if mode == "fan_in" else fan_out
    gain = nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device="cpu"):
    is_np = isinstance(coordinates, np.ndarray)
    c = coordinates if is_np else coordinates.numpy()
    if quantization_size is not None:
        q = np.asarray(quantization_size, dtype=c.dtype if c.dtype.kind == "f" else np.float64)
        c = np.floor(c / q)
    vox = np.ascontiguousarray(c.astype(np.int32))
    n = vox.shape[0]
    index = np.empty(n, dtype=np.int32)
    inverse = np.empty(n, dtype=np.int32)
    lab = None
    vlab = None
    if labels is not None:
        lab = np.ascontiguousarray(np.asarray(labels).astype(np.int32))
        vlab = np.empty(n, dtype=np.int32)
    m = lib().orc_sparse_quantize(vox.ctypes.data, n, lab.ctypes.data if lab is not None else None,
                                  int(ignore_label), index.ctypes.data, inverse.ctypes.data,
                                  vlab.ctypes.data if vlab is not None else None)
    index, inverse = index[:m].astype(np.int64), inverse.astype(np.int64)
    conv = (lambda a: a) if is_np else torch.from_numpy
    if return_maps_only:
        return (conv(index), conv(inverse)) if return_inverse else conv(index)
    ret = [conv(vox[index])]
    if features is not None:
        ret.append(features[index] if isinstance(features, np.ndarray) else features[torch.from_numpy(index)])
    if labels is not None:
        ret.append(conv(vlab[:m].astype(np.asarray(labels).dtype)))
    if return_index:
        ret.append(conv(index))
    if return_inverse:
        ret.append(conv(inverse))
    return ret[0] if len(ret) == 1 else tuple(ret)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        bc = torch.full((c.shape[0], 1), b, dtype=c.dtype)
        out.append(torch.cat([bc, c], dim=1))
    return torch.cat(out, dim=0).to(dtype)


class SparseCollation:
    def __init__(self, limit_numpoints=-1, dtype=torch.int32, device=None):
        self.dtype, self.device = dtype, device

    def __call__(self, list_data):
        coords, feats, labels = list(zip(*list_data))
        coords_batch = batched_coordinates(coords, dtype=self.dtype)
        feats_batch = torch.cat([torch.as_tensor(f) for f in feats], dim=0)
        labels_batch = torch.cat([torch.as_tensor(l) for l in labels], dim=0)
        return coords_batch, feats_batch, labels_batch


utils = types.ModuleType(__name__ + ".utils")
utils.kaiming_normal_ = kaiming_normal_
utils.sparse_quantize = sparse_quantize
utils.SparseCollation = SparseCollation
utils.batched_coordinates = batched_coordinates


# ---------------------------------------------------------------- modules.resnet_block
class BasicBlock(nn.Module):
    """ME's BasicBlock (structure evidenced in the reference by utils/models/resnet_block.py:8-56)."""
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
        residual = x
        out = self.conv1(x)
        out = self.norm1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.norm2(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        out = self.relu(out)
        return out


class Bottleneck(nn.Module):
    """ME's Bottleneck block (structure evidenced in the reference by utils/models/resnet_block.py:59-117)."""
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
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.relu(self.norm2(self.conv2(out)))
        out = self.norm3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


modules = types.ModuleType(__name__ + ".modules")
resnet_block = types.ModuleType(__name__ + ".modules.resnet_block")
resnet_block.BasicBlock = BasicBlock
resnet_block.Bottleneck = Bottleneck
modules.resnet_block = resnet_block


def install_as_minkowski_engine():
    """Alias this oracle as ``MinkowskiEngine`` so the reference's model files import (container only)."""
    me = sys.modules[__name__]
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.utils"] = utils
    sys.modules["MinkowskiEngine.modules"] = modules
    sys.modules["MinkowskiEngine.modules.resnet_block"] = resnet_block
    return me
