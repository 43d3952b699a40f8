"""MinkUNet34 / MinkUNet34BEV wiring, generic over the operator backend.

Mirrors the module names, parameter names (state_dict keys) and call order of
the reference models so reference checkpoints load unchanged:

  MinkUNet34BEV  utils/models/minkunet_bev.py:9-156 (layers), :302-399 (forward), :445-447
  MinkUNet34     utils/models/minkunet.py:23-95 (layers), :97-158 (forward), :171-174
  BasicBlock     MinkowskiEngine.modules.resnet_block (evidence: utils/models/resnet_block.py:8-56)

``make_models(ME, Encoder2D, sparse2super)`` binds the wiring to an operator
module exposing the MinkowskiEngine names of SURVEY.md 8(b).  The product binds
it to ``lidog_amd.me`` (HIP kernels) at the bottom of ``lidog_amd/__init__.py``;
the test-suite binds the same wiring to the CPU oracle, after proving in the
build container that this wiring equals the reference classes bit for bit.
"""
import types

import torch.nn as nn

# (PLANES, LAYERS) of MinkUNet34 (minkunet_bev.py:14,445-447)
PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
LAYERS34 = (2, 3, 4, 6, 2, 2, 2, 2)
INIT_DIM = 32
# encoder stage i: conv{i}p{s}s2 / bn{i} / block{i};  decoder stage j: convtr{j}p{s}s2 / bntr{j} / block{j+1}
_ENC = [(1, 1), (2, 2), (3, 4), (4, 8)]
_DEC = [(4, 16), (5, 8), (6, 4), (7, 2)]
BEV_LEVEL_CHANNELS = {"block8": 96, "block7": 96, "block6": 128, "bottle": 256}


def make_models(ME, Encoder2D=None, sparse2super=None):
    BasicBlock = ME.modules.resnet_block.BasicBlock
    _fused = getattr(ME, "bn_relu", None)  # optional backend fast path: BN + ReLU in one kernel
    _conv_bn = getattr(ME, "conv_bn", None)
    _trunk_exec = getattr(ME, "trunk_forward", None)  # optional: the whole trunk as one launch sequence

    class _Trunk(nn.Module):
        BLOCK = BasicBlock
        LAYERS = LAYERS34

        def _build_trunk(self, in_channels, out_channels, D, initial_kernel_size):
            self.D = D
            self.inplanes = INIT_DIM
            self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, INIT_DIM, kernel_size=initial_kernel_size,
                                                     dimension=D)
            self.bn0 = ME.MinkowskiBatchNorm(INIT_DIM)
            for i, s in _ENC:
                setattr(self, f"conv{i}p{s}s2", ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2,
                                                                       stride=2, dimension=D))
                setattr(self, f"bn{i}", ME.MinkowskiBatchNorm(self.inplanes))
                setattr(self, f"block{i}", self._make_layer(PLANES[i - 1], self.LAYERS[i - 1]))
            skips = [PLANES[2], PLANES[1], PLANES[0], INIT_DIM]
            for (j, s), skip in zip(_DEC, skips):
                setattr(self, f"convtr{j}p{s}s2", ME.MinkowskiConvolutionTranspose(self.inplanes, PLANES[j],
                                                                                  kernel_size=2, stride=2,
                                                                                  dimension=D))
                setattr(self, f"bntr{j}", ME.MinkowskiBatchNorm(PLANES[j]))
                self.inplanes = PLANES[j] + skip
                setattr(self, f"block{j + 1}", self._make_layer(PLANES[j], self.LAYERS[j]))
            self.final = ME.MinkowskiConvolution(PLANES[7], out_channels, kernel_size=1, bias=True, dimension=D)
            self.relu = ME.MinkowskiReLU(inplace=True)
            self.dropout = ME.MinkowskiDropout(p=0.5)  # constructed, never called (minkunet_bev.py:126)

        def _make_layer(self, planes, blocks):
            downsample = None
            if self.inplanes != planes:
                downsample = nn.Sequential(
                    ME.MinkowskiConvolution(self.inplanes, planes, kernel_size=1, stride=1, dimension=self.D),
                    ME.MinkowskiBatchNorm(planes))
            layers = [self.BLOCK(self.inplanes, planes, stride=1, dilation=1, downsample=downsample,
                                 dimension=self.D)]
            self.inplanes = planes
            for _ in range(1, blocks):
                layers.append(self.BLOCK(self.inplanes, planes, stride=1, dilation=1, dimension=self.D))
            return nn.Sequential(*layers)

        def weight_initialization(self):
            # minkunet_bev.py:401-408: only MinkowskiConvolution (not ...Transpose) gets kaiming fan_out
            for m in self.modules():
                if isinstance(m, ME.MinkowskiConvolution):
                    ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
                if isinstance(m, ME.MinkowskiBatchNorm):
                    nn.init.constant_(m.bn.weight, 1)
                    nn.init.constant_(m.bn.bias, 0)

        def _bn_relu(self, bn, x):
            return _fused(bn, x) if _fused is not None else self.relu(bn(x))

        def _conv_bn_relu(self, conv, bn, x):
            if _conv_bn is not None:  # backend fast path: BN statistics from the convolution's own epilogue
                return _conv_bn(conv, bn, x, relu=True)
            return self._bn_relu(bn, conv(x))

        def _trunk_forward(self, x):
            """returns (out_block8, out_bottle, {level: tensor}, classifier output or None = not computed yet)"""
            if _trunk_exec is not None:
                done = _trunk_exec(self, x)
                if done is not None:
                    return done
            out = self._conv_bn_relu(self.conv0p1s1, self.bn0, x)
            skips = [out]
            for i, s in _ENC:
                out = self._conv_bn_relu(getattr(self, f"conv{i}p{s}s2"), getattr(self, f"bn{i}"), out)
                out = getattr(self, f"block{i}")(out)
                skips.append(out)
            bottle = skips.pop()
            levels = {}
            names = ["bottle", "block6", "block7", "block8"]
            for (j, s), name in zip(_DEC, names):
                out = self._conv_bn_relu(getattr(self, f"convtr{j}p{s}s2"), getattr(self, f"bntr{j}"), out)
                out = ME.cat(out, skips.pop())
                out = getattr(self, f"block{j + 1}")(out)
                levels[name] = out
            return out, bottle, levels, None

    class MinkUNet34(_Trunk):
        def __init__(self, in_channels, out_channels, D=3, initial_kernel_size=5):
            super().__init__()
            self._build_trunk(in_channels, out_channels, D, initial_kernel_size)
            self.weight_initialization()

        def forward(self, x, is_seg=True):
            out, _, _, seg = self._trunk_forward(x)
            seg = seg if seg is not None else self.final(out)
            return seg if is_seg else (seg, out)

    class MinkUNet34BEV(_Trunk):
        def __init__(self, in_channels, out_channels, D, initial_kernel_size=5, dynamic_mapping=False,
                     decoder_2d_level=("block8",), bottle_img_dim=None, bottle_out_img_dim=None,
                     mapping_bound_2d=50.0, scaling_factors=None, binary_seg_layer=False):
            super().__init__()
            assert not binary_seg_layer, "binary_seg_layer is off in every LiDOG config"
            self.mapping_bound_2d = mapping_bound_2d
            self.decoder_2d_level = list(decoder_2d_level)
            self.scaling_factors = scaling_factors or {k: 1.0 for k in BEV_LEVEL_CHANNELS}
            self._build_trunk(in_channels, out_channels, D, initial_kernel_size)
            self.encoders2d = nn.ModuleDict({k: Encoder2D(BEV_LEVEL_CHANNELS[k], n_classes=out_channels)
                                             for k in self.decoder_2d_level})
            self.weight_initialization()

        def forward(self, x, is_seg=True, is_train=False):
            out, bottle, levels, seg = self._trunk_forward(x)
            img_pred = None
            if is_train:
                img_pred = {}
                for key in self.encoders2d.keys():
                    stride = int(3 / self.scaling_factors[key])
                    bev = sparse2super(levels[key], bound=self.mapping_bound_2d, voxel=0.05, pool=(5, stride, 1))
                    img_pred[key] = self.encoders2d[key](bev)
            seg = seg if seg is not None else self.final(out)
            if is_seg:
                return seg, img_pred
            return seg, img_pred, bottle, None

    return types.SimpleNamespace(MinkUNet34=MinkUNet34, MinkUNet34BEV=MinkUNet34BEV)
